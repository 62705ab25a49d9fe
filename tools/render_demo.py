"""Renders demo images through the C ABI and writes PNGs: Cornell box (path, 256 spp, Gaussian filter), the Cornell
box with spheres (mirror / glass / emitter), the config-3 cloud. Usage: python tools/render_demo.py [out_dir]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np, pbrt_hip
from pbrt_hip import scenes

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
ctx = pbrt_hip.Context(0)
gauss = pbrt_hip.filter_table("gaussian", 1.5, 1.5, 2.0, 0.0)


def save(name, film, exposure=1.0):
    rgb = pbrt_hip.film_to_rgb(film) * exposure
    pbrt_hip.write_png(os.path.join(out, name + ".png"), rgb)
    print(name, "mean", float(rgb.mean()), flush=True)


w = h = 384
g = pbrt_hip.Scene(ctx, scenes.cornell_box())
film, st = g.render(scenes.cornell_camera(w, h), w, h, 256, max_depth=8, light_strategy=1, seed=0, filter=gauss,
                    sampler=("zerotwo", 4))
save("demo_cornell", film, 0.6)
g.close()

sc = scenes.cornell_box()
sc["materials"] = np.concatenate([sc["materials"], scenes._materials([(scenes.MAT_MIRROR, (0.9, 0.9, 0.9), (0, 0, 0), 1.0),
                                                                       (scenes.MAT_GLASS, (1, 1, 1), (1, 1, 1), 1.5)])])
sph = np.zeros((2, 8), dtype=np.float32)
sph[0] = (400.0, 420.0, 300.0, 70.0, 3, -1, 0, 0)     # mirror ball
sph[1] = (190.0, 235.0, 170.0, 70.0, 4, -1, 0, 0)     # glass ball on the short box
sc["spheres"] = sph
g = pbrt_hip.Scene(ctx, sc)
film, st = g.render(scenes.cornell_camera(w, h), w, h, 256, max_depth=12, light_strategy=1, seed=1, filter=gauss,
                    sampler=("halton",))
save("demo_cornell_spheres", film, 0.6)
g.close()

w, h = 480, 270
g = pbrt_hip.Scene(ctx, scenes.random_triangles(1_000_000, seq=1), device_build=True)
film, st = g.render(scenes.random_triangles_camera(w, h), w, h, 64, max_depth=5, seed=0)
save("demo_config3", film)
g.close()
ctx.close()
