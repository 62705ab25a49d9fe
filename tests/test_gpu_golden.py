"""GPU path against the committed fixtures (tests/golden/): the oracle is not consulted here."""
import os

import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hit_table(hip_ctx):
    d = np.load(os.path.join(G, "hit_table_1k.npz"))
    sc = scenes.random_triangles(1000, seq=42, size=0.15)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    assert gsc.nodes.tobytes() == d["nodes"].tobytes()
    hits = gsc.intersect(d["rays"])
    exp = d["hits"]
    assert np.array_equal(hits["prim_id"], exp["prim_id"])
    m = exp["prim_id"] >= 0
    for f in ("t", "b0", "b1", "b2"):
        assert np.array_equal(hits[f][m].view(np.uint32), exp[f][m].view(np.uint32))
    assert np.array_equal(gsc.intersect_p(d["rays"]), d["occluded"])
    hip_ctx.set_counting(True)
    try:
        hip_ctx.counters(reset=True)
        gsc.intersect(d["rays"])
        c0 = hip_ctx.counters(reset=True)
        gsc.intersect_p(d["rays"])
        c1 = hip_ctx.counters(reset=True)
    finally:
        hip_ctx.set_counting(False)
    assert [c0["rays"], c0["node_tests"], c0["prim_tests"], c1["rays"], c1["node_tests"], c1["prim_tests"]] == \
        d["counters"].tolist()
    gsc.close()


@pytest.mark.parametrize("name", ["cornell_64x64x4.npz", "mixed_64x64x4.npz"])
def test_images(hip_ctx, name):
    w = h = 64
    d = np.load(os.path.join(G, name))
    if name.startswith("cornell"):
        sc, cam, kw = scenes.cornell_box(), scenes.cornell_camera(w, h), dict(max_depth=8, seed=0)
    else:
        sc, cam, kw = scenes.mixed_materials_scene(), scenes.random_triangles_camera(w, h), dict(max_depth=16, seed=5)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    film, st = gsc.render(cam, w, h, 4, **kw)
    exp = d["film"]
    assert np.array_equal(film[..., 3], exp[..., 3])
    rgb, rgb_e = pbrt_hip.film_to_rgb(film), pbrt_hip.film_to_rgb(exp)
    assert np.all(np.abs(rgb - rgb_e) <= 1e-5 * np.maximum(1.0, np.abs(rgb_e)))
    assert float(np.sqrt(np.mean((rgb.astype(np.float64) - rgb_e) ** 2))) <= 1e-6   # north_star budget: 1e-4
    assert st["rays_closest"] + st["rays_shadow"] == int(d["stats"][0])
    assert st["camera_samples"] == int(d["stats"][3])
    gsc.close()


def test_thin_lens_camera(hip_ctx):
    """PerspectiveCamera with lens_radius > 0 (perspective.rs:99-105) vs the oracle."""
    import oracle
    w, h = 64, 48
    sc = scenes.cornell_box()
    cam = scenes.perspective_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), (0, 1, 0), 39.3, w, h,
                                    lens_radius=20.0, focal_distance=1000.0)
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    fc, _ = osc.render(scenes.camera_dict_to_floats(cam), w, h, 8, max_depth=3, seed=9)
    fg, _ = gsc.render(cam, w, h, 8, max_depth=3, seed=9)
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(fg), oracle.film_to_rgb(fc)
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    gsc.close()
    osc.close()


def test_variant_fixtures(hip_ctx):
    """The device against the committed fixture of every widened row (tests/golden/variants.py)."""
    import sys
    sys.path.insert(0, G)
    from variants import W, H, SPP, variants
    d = np.load(os.path.join(G, "variants_48x32.npz"))
    for name, (sc, cam, kw) in variants().items():
        kw = dict(kw)
        spec = kw.pop("filter_spec", None)
        if spec is not None:
            kw["filter"] = pbrt_hip.filter_table(spec[0], spec[1], spec[1], spec[2], spec[3])
        gsc = pbrt_hip.Scene(hip_ctx, sc)
        film, st = gsc.render(cam, W, H, SPP, **kw)
        gsc.close()
        exp = d[name]
        if spec is None:
            assert np.array_equal(film[..., 3], exp[..., 3]), name
        else:
            assert np.allclose(film[..., 3], exp[..., 3], rtol=2e-5, atol=2e-5), name
        rgb, rgb_e = pbrt_hip.film_to_rgb(film), pbrt_hip.film_to_rgb(exp)
        assert np.all(np.abs(rgb - rgb_e) <= 2e-5 * np.maximum(1.0, np.abs(rgb_e))), name
        assert st["rays_closest"] + st["rays_shadow"] == int(d[name + "__rays"][0]), name
        assert st["camera_samples"] == int(d[name + "__rays"][1]), name
