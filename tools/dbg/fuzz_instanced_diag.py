"""Which ingredient of the instanced fuzz_render cases makes GPU and oracle disagree? Re-runs failing seeds with one ingredient
neutralised at a time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle, pbrt_hip, fuzz_render
from pbrt_hip import scenes
ctx = pbrt_hip.Context(0)

def run(sc, cam, w, h, spp, kw, g):
    osc = oracle.OracleScene(sc); gsc = pbrt_hip.Scene(ctx, sc)
    fc, stc = osc.render(scenes.camera_dict_to_floats(cam), w, h, spp, n_threads=4, **kw)
    fg, stg = gsc.render(cam, w, h, spp, **kw, **g)
    gsc.close(); osc.close()
    a, b = pbrt_hip.film_to_rgb(fg), oracle.film_to_rgb(fc)
    off = int((np.abs(a - b) > 1e-5 * np.maximum(1, np.abs(b))).any(axis=-1).sum())
    return off, stg["rays_closest"] + stg["rays_shadow"], stc["rays"]

for seed in [int(x) for x in sys.argv[1:]] or [200708, 200830, 201385, 201473, 200182]:
    sc, cam, w, h, spp, kw, g, desc = fuzz_render.make_case(seed)
    print("seed", seed, desc[:150])
    print("   as is:", run(sc, cam, w, h, spp, kw, g), "instance_material", sc["instance_material"], "tri_material kinds", np.unique(sc["tri_material"]))
    v = dict(sc); v["instance_material"] = np.zeros_like(sc["instance_material"]); print("   all instances matte override:", run(v, cam, w, h, spp, kw, g))
    v = dict(sc); v["instance_material"] = np.full_like(sc["instance_material"], -1); print("   no overrides:", run(v, cam, w, h, spp, kw, g))
    v = dict(sc); v["tri_material"] = np.zeros_like(sc["tri_material"]); print("   all triangles matte:", run(v, cam, w, h, spp, kw, g))
    v = dict(sc); v["tri_material"] = np.zeros_like(sc["tri_material"]); v["instance_material"] = np.full_like(sc["instance_material"], -1); print("   all matte, no overrides:", run(v, cam, w, h, spp, kw, g))
    v = dict(sc); v["lights"] = scenes._lights([(scenes.LIGHT_INFINITE, (1.0, 1.0, 1.0), -1, 0, 1)]); print("   env light only:", run(v, cam, w, h, spp, kw, g))
    print("   shade_order 0, one pass:", run(sc, cam, w, h, spp, kw, dict(shade_order=0, spp_per_pass=0)))
    k2 = dict(kw); k2.pop("sampler", None); print("   random sampler:", run(sc, cam, w, h, spp, k2, g))
    k2 = dict(kw); k2["max_depth"] = 1; print("   max_depth 1:", run(sc, cam, w, h, spp, k2, g))
    cam2 = dict(cam) if isinstance(cam, dict) else cam
