import os, sys, time
sys.path.insert(0, "/root/repo/pbrt-rs_amd")
import pbrt_hip
from pbrt_hip import scenes
W, H, spp = 1920, 1080, 16
sc = scenes.random_triangles(1_000_000, seq=1)
cam = scenes.random_triangles_camera(W, H)
bvh = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, 0)
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc, bvh=bvh)
out = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    film, st = g.render(cam, W, H, spp, max_depth=5, seed=0)
    out.append(f"{st['total_ms']:.0f}/{st['trace_ms']:.0f}")
print("total/trace ms per frame:", " ".join(out), flush=True)
