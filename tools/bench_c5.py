"""Config 5 (10 000 base triangles x 1000 rigid instances = 10 M instanced, matte/mirror/glass by instance,
env light, PathIntegrator depth 16) at a chosen resolution / spp on one GPU. Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
from pbrt_hip import scenes
W, H, spp = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160)), int(os.environ.get("SPP", 4))
t = time.time()
sc = scenes.instanced_scene(10_000, 1000)
bvh = pbrt_hip.build_two_level(sc)
print(f"scene + host BVHs {time.time()-t:.2f} s; tlas nodes {len(bvh[3])} blas nodes {len(bvh[0])}")
cam = scenes.instanced_camera(W, H)
ctx = pbrt_hip.Context(0)
scene = pbrt_hip.Scene(ctx, sc, bvh=bvh)
for it in range(2):
    film, st = scene.render(cam, W, H, spp, max_depth=16, seed=0)
rays = st["rays_closest"] + st["rays_shadow"]
print(f"{W}x{H}x{spp}: total {st['total_ms']:.1f} ms trace {st['trace_ms']:.1f} ms ({st['trace_launches']} launches) "
      f"rays {rays/1e6:.1f}M -> {rays/st['total_ms']/1e3:.0f} Mrays/s")
