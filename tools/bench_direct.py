import os, sys
sys.path.insert(0, "/root/repo/pbrt-rs_amd")
import pbrt_hip
pbrt_hip.LIB_PATH = sys.argv[1]
from pbrt_hip import scenes
w, h, spp = 1920, 1080, 16
sc = scenes.random_triangles(1_000_000, seq=1)
cam = scenes.random_triangles_camera(w, h)
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc)
for label, kw in (("direct", dict(integrator=1, max_depth=5, light_strategy=1)), ("whitted", dict(integrator=2, max_depth=5)),
                  ("ao16", dict(integrator=3, ao_samples=16))):
    g.render(cam, w, h, 2, seed=0, **kw)
    film, st = g.render(cam, w, h, spp, seed=0, **kw)
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"{os.path.basename(sys.argv[1])} {label}: {st['total_ms']:.1f} ms (trace {st['trace_ms']:.1f}) {rays / st['total_ms'] / 1e3:.0f} Mrays/s", flush=True)
