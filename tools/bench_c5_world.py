"""Config 5 with the instances written out as world-space triangles (scenes.world_space_instances): 10 M real
triangles in one BVH instead of 1000 TransformedPrimitives of a 10 k mesh. Usage: W=1920 H=1080 SPP=8 python ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np, pbrt_hip
from pbrt_hip import scenes
W, H, spp = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), int(os.environ.get("SPP", 8))
sc = scenes.instanced_scene(10_000, 1000)
cam = scenes.instanced_camera(W, H)
ctx = pbrt_hip.Context(0)
t = time.time()
flat = scenes.world_space_instances(sc)
print(f"world-space mesh: {flat['indices'].shape[0]} triangles in {time.time() - t:.2f} s", flush=True)
for label, kw in (("device HLBVH", dict(device_build=True)), ("host SAH", dict())):
    t = time.time()
    g = pbrt_hip.Scene(ctx, flat, **kw)
    setup = time.time() - t
    for it in range(2):
        film, st = g.render(cam, W, H, spp, max_depth=16, seed=0)
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"{label:13s} setup {setup:6.2f} s | {W}x{H}x{spp}: {st['total_ms']:.1f} ms, {rays / 1e6:.1f} M rays -> "
          f"{rays / st['total_ms'] / 1e3:.0f} Mrays/s (trace {st['trace_ms'] / st['total_ms']:.2f} of frame)", flush=True)
    g.close()
bvh = pbrt_hip.build_two_level(sc)
g = pbrt_hip.Scene(ctx, sc, bvh=bvh)
for it in range(2):
    film, st = g.render(cam, W, H, spp, max_depth=16, seed=0)
rays = st["rays_closest"] + st["rays_shadow"]
print(f"two-level (TransformedPrimitive)      | {W}x{H}x{spp}: {st['total_ms']:.1f} ms, {rays / 1e6:.1f} M rays -> "
      f"{rays / st['total_ms'] / 1e3:.0f} Mrays/s", flush=True)
