"""Config 5 (10 000 base triangles x 1000 rigid instances = 10 M instanced, matte/mirror/glass by instance,
env light, PathIntegrator depth 16) at a chosen resolution / spp on one GPU. Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
if os.environ.get("PBRT_LIB"): pbrt_hip.LIB_PATH = os.environ["PBRT_LIB"]
from pbrt_hip import scenes
W, H, spp = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160)), int(os.environ.get("SPP", 4))
ORDER = int(os.environ.get("SHADE_ORDER", "0"))   # PbrtRenderParams.shade_order
t = time.time()
sc = scenes.instanced_scene(10_000, 1000)
bvh = pbrt_hip.build_two_level(sc, tlas_max_prims=int(os.environ.get("TLAS_MAX_PRIMS", "4")))
print(f"scene + host BVHs {time.time()-t:.2f} s; tlas nodes {len(bvh[3])} blas nodes {len(bvh[0])}")
cam = scenes.instanced_camera(W, H)
ctx = pbrt_hip.Context(0)
scene = pbrt_hip.Scene(ctx, sc, bvh=bvh)
print("wide records:", scene.wide_records())
ctx.set_traversal(int(os.environ.get("TRAVERSAL", "0")))   # 3 = PBRT_TRAVERSAL_ROUNDS
for it in range(2):
    film, st = scene.render(cam, W, H, spp, max_depth=16, seed=0, shade_order=ORDER, samples_per_wave=int(os.environ.get("SPW", "0")))
rays = st["rays_closest"] + st["rays_shadow"]
print(f"{W}x{H}x{spp}: total {st['total_ms']:.1f} ms trace {st['trace_ms']:.1f} ms ({st['trace_launches']} launches) "
      f"rays {rays/1e6:.1f}M -> {rays/st['total_ms']/1e3:.0f} Mrays/s")
if os.environ.get("NO_COUNT"): sys.exit(0)
if scene.wide_records()[0] > 0:
    ctx.set_counting(2); ctx.wide_counters(reset=True)
    scene.render(cam, W, H, spp, max_depth=16, seed=0, shade_order=ORDER, samples_per_wave=int(os.environ.get("SPW", "0")))
    wc = ctx.wide_counters(reset=True)
    print("wide per ray: " + " ".join(f"{k} {v/rays:.2f}" for k, v in wc.items()))
ctx.set_counting(True); ctx.counters(reset=True)
film, st2 = scene.render(cam, W, H, spp, max_depth=16, seed=0, shade_order=ORDER, samples_per_wave=int(os.environ.get("SPW", "0")))
c = ctx.counters(reset=True); ctx.set_counting(False)
rays = c["rays"]
alg = 32*rays + 32*c["node_tests"] + 48*c["prim_tests"] + 112*c.get("inst_tests",0) + 16*st2["rays_closest"] + 4*st2["rays_shadow"]
print(f"per ray: node {c['node_tests']/rays:.1f} tri {c['prim_tests']/rays:.2f} inst {c.get('inst_tests',0)/rays:.2f}; "
      f"algorithmic {alg/rays:.0f} B/ray -> {alg/st['trace_ms']/1e6:.0f} GB/s ({alg/st['trace_ms']/1e6/8000:.2f} of 8 TB/s)")
