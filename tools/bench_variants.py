"""Throughput of the widened rows on the config-3 scene (1 M triangles, 1920x1080): samplers, sibling integrators,
delta lights, device-built HLBVH. Scene resident, film on the device.
Usage: python tools/bench_variants.py [n_tris] [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np, pbrt_hip
from pbrt_hip import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, h = 1920, 1080
sc = scenes.random_triangles(n, seq=1)
cam = scenes.random_triangles_camera(w, h)
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc)


def run(label, scene=g, **kw):
    scene.render(cam, w, h, 4 if "sampler" not in kw else spp, **kw)      # warm up
    film, st = scene.render(cam, w, h, spp, **kw)
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"{label:44s} {rays / st['total_ms'] * 1e-3:8.1f} Mrays/s  frame {st['total_ms']:8.1f} ms  "
          f"rays {rays / 1e6:8.1f} M  trace {st['trace_ms'] / st['total_ms']:.2f} of frame", flush=True)


run("path, random sampler (bench.py config)", max_depth=5, seed=0)
run("path, stratified 8x8, 4 dims", max_depth=5, seed=0, sampler=("stratified", 8, 8, True, 4))
run("path, (0,2)-sequence, 4 dims", max_depth=5, seed=0, sampler=("zerotwo", 4))
run("path, Halton", max_depth=5, seed=0, sampler=("halton",))
run("direct lighting (one light)", integrator=1, max_depth=5, light_strategy=1, seed=0)
run("Whitted", integrator=2, max_depth=5, seed=0)
run("ambient occlusion, 16 cosine samples", integrator=3, ao_samples=16, cos_sample=True, seed=0)
g2 = pbrt_hip.Scene(ctx, scenes.with_lights(sc, [scenes.point_light((0.0, 3.0, 0.0), (20.0, 20.0, 20.0)),
                                                   scenes.distant_light((0.2, 1.0, 0.1), (1.0, 1.0, 1.0))]))
run("path, env + point + distant lights (power)", scene=g2, max_depth=5, light_strategy=1, seed=0)
g2.close()
g3 = pbrt_hip.Scene(ctx, sc, device_build=True)
run(f"path over device-built HLBVH (build {g3.build_ms:.1f} ms)", scene=g3, max_depth=5, seed=0)
g3.close()
g.close()
ctx.close()
