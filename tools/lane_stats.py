"""Lane-utilisation breakdown of the traversal kernel (needs a library built with PB_DEFS=-DPB_LANE_STATS):
    PB_DEFS=-DPB_LANE_STATS PB_OUT=/tmp/libpbrt_stats.so pbrt-rs_amd/build.sh; python tools/lane_stats.py /tmp/libpbrt_stats.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np, pbrt_hip
pbrt_hip.LIB_PATH = sys.argv[1]
from pbrt_hip import scenes
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "8"))
bvh = None
if os.environ.get("INSTANCED"):
    sc, cam, depth = scenes.instanced_scene(10_000, 1000), scenes.instanced_camera(W, H), 16
    bvh = pbrt_hip.build_two_level(sc, tlas_max_prims=int(os.environ.get("TLAS_MAX_PRIMS", "4")))
else:
    sc, cam, depth = scenes.random_triangles(int(os.environ.get("TRIS", "1000000")), seq=1), scenes.random_triangles_camera(W, H), 5
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc, bvh=bvh)
g.render(cam, W, H, spp, max_depth=depth, seed=0)
L = pbrt_hip.lib()
out = (ctypes.c_uint64 * 8)()
wout = (ctypes.c_uint64 * 32)()
L.pbrt_hip_debug_lane_stats(out, 1)
L.pbrt_hip_debug_wide_stats(wout, 1)
film, st = g.render(cam, W, H, spp, max_depth=depth, seed=0)
L.pbrt_hip_debug_lane_stats(out, 1)
L.pbrt_hip_debug_wide_stats(wout, 1)
s = [int(v) for v in out]
w = [int(v) for v in wout]
rays = st["rays_closest"] + st["rays_shadow"]
print(f"rays {rays/1e6:.1f} M, trace {st['trace_ms']:.1f} ms, wide records {g.wide_records()}")
if w[0]:
    print(f"wide records: {w[0]/1e6:.1f} M wave iterations, {w[1]/w[0]:.1f} lanes active of 64 ({w[11]/rays:.1f} record steps per ray), "
          f"{w[8]/max(w[11],1):.2f} children pass the filter per step")
    print(f"wide leaves:  {w[2]/1e6:.1f} M wave sections, {w[3]/max(w[2],1):.1f} lanes with a candidate leaf ({w[12]/rays:.2f} candidates per ray), "
          f"{w[4]/max(w[12],1):.2f} of them pass the exact box, {w[5]/rays:.2f} triangle tests per ray")
    print(f"wide outer:   {w[6]/1e6:.1f} M iterations, {w[7]/max(w[6],1):.1f} lanes with work")
    cyc = w[13] + w[14] + w[15]
    if cyc:
        print(f"wide refills: {w[9]/1e6:.2f} M ({w[9]/max(w[6],1):.2f} per outer iteration), {w[10]/max(w[9],1):.1f} rays fetched per refill")
        print(f"wave time (s_memtime between sections): refill {100*w[13]/cyc:.1f} %, record loop {100*w[14]/cyc:.1f} %, leaf phase {100*w[15]/cyc:.1f} %")
    if w[21]:
        print(f"two-level per ray: top-leaf box tests {w[16]/rays:.2f} (pass {w[17]/max(w[16],1):.2f}), entry attempts {w[18]/rays:.2f}, "
              f"entered {w[19]/rays:.2f}, exit turns {w[20]/rays:.2f}")
        print(f"instance sections: {w[21]/1e6:.1f} M at {w[22]/w[21]:.1f} lanes, {w[23]/w[21]:.2f} entry-loop trips per section")
    if w[26]:
        cyc = max(w[13] + w[14] + w[15], 1)
        print(f"triangle sections: {w[26]/1e6:.1f} M at {w[28]/w[26]:.1f} lanes, {w[27]/w[26]:.2f} loop trips per section")
        print(f"leaf phase split: instance branch {100*w[24]/cyc:.1f} % of wave time, triangle branch {100*w[25]/cyc:.1f} %")
    sys.exit(0)
print(f"interior: {s[0]/1e6:.1f} M wave iterations, {s[1]/max(s[0],1):.1f} lanes active of 64 ({s[1]/rays:.1f} node steps per ray)")
print(f"leaves:   {s[2]/1e6:.1f} M wave sections, {s[3]/max(s[2],1):.1f} lanes with a leaf, {s[4]/max(s[2],1):.2f} loop trips per section, "
      f"{s[5]/max(s[3],1):.2f} triangles per lane-leaf, {s[5]/rays:.2f} tri tests per ray; lane-trip utilisation {s[5]/max(s[4]*64,1):.2f}")
print(f"outer:    {s[6]/1e6:.1f} M iterations, {s[7]/max(s[6],1):.1f} lanes with work")
print(f"wave-instruction slots: interior {s[0]/1e6:.1f} M vs leaf trips {s[4]/1e6:.1f} M")
