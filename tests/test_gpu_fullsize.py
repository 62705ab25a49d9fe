"""Parity and size-independent properties at BASELINE.json's sizes:
config 2 (Cornell 512x512x64 spp, depth 8) against the oracle in full; config 3 (1 M triangles,
1920x1080) against the oracle on a 256x256 crop at 64 spp, plus properties that need no oracle at
full resolution: determinism, tile-partition linearity, closest-hit / any-hit consistency, t_max
monotonicity."""
import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def test_config2_cornell_full(hip_ctx):
    w = h = 512
    sc, cam = scenes.cornell_box(), scenes.cornell_camera(512, 512)
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    kw = dict(max_depth=8, rr_threshold=1.0, light_strategy=1, seed=0)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, 64, n_threads=16, **kw)
    film_g, st_g = gsc.render(cam, w, h, 64, **kw)
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g), oracle.film_to_rgb(film_c)
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))   # last-bit float differences only
    assert _rmse(rgb_g, rgb_c) <= 1e-5                      # north_star budget: 1e-4 (emitter pixels are ~17)
    gsc.close()
    osc.close()


@pytest.fixture(scope="module")
def config3(hip_ctx):
    sc = scenes.random_triangles(1_000_000, seq=1)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    yield sc, gsc
    gsc.close()


def test_config3_crop_vs_oracle(hip_ctx, config3):
    sc, gsc = config3
    w, h = 1920, 1080
    cam = scenes.random_triangles_camera(w, h)
    bounds = (832, 412, 1088, 668)                          # 256x256 centre crop
    osc = oracle.OracleScene(sc)
    assert osc.nodes().tobytes() == gsc.nodes.tobytes()     # 1 M-triangle SAH tree: host builder == oracle
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, 64, max_depth=5, seed=0, bounds=bounds,
                              n_threads=16)
    film_g, st_g = gsc.render(cam, w, h, 64, max_depth=5, seed=0, bounds=bounds)
    osc.close()
    crop = (slice(bounds[1], bounds[3]), slice(bounds[0], bounds[2]))
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g[crop]), oracle.film_to_rgb(film_c[crop])
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    assert _rmse(rgb_g, rgb_c) <= 1e-6
    assert np.all(film_g[:bounds[1]] == 0)


def test_config3_full_frame_properties(hip_ctx, config3):
    sc, gsc = config3
    w, h, spp = 1920, 1080, 4
    cam = scenes.random_triangles_camera(w, h)
    a, st_a = gsc.render(cam, w, h, spp, max_depth=5, seed=0)
    b, st_b = gsc.render(cam, w, h, spp, max_depth=5, seed=0, spp_per_pass=3)
    assert a.tobytes() == b.tobytes()                        # deterministic, independent of the pass split
    assert st_a["rays_closest"] == st_b["rays_closest"] and st_a["rays_shadow"] == st_b["rays_shadow"]
    # every pixel received its spp unit filter weights; a few also one from a neighbour whose film position
    # x + u rounded to a whole pixel (add_sample's ceil / floor, film.rs:263-264)
    assert np.all(a[..., 3] >= spp) and (a[..., 3] != spp).mean() < 1e-3
    parts = [gsc.render(cam, w, h, spp, max_depth=5, seed=0, tile_rank=r, tile_world=4)[0] for r in range(4)]
    owned = sum((p[..., 3] >= spp).astype(np.int32) for p in parts)
    assert np.all(owned == 1)                                # the four ranks' tile sets partition the frame
    total = sum(parts)
    plain = a[..., 3] == spp
    assert np.array_equal(total[plain], a[plain])            # and their films sum to it exactly
    assert np.allclose(total, a, rtol=1e-6, atol=1e-6)       # (neighbour contributions: float atomics, any order)
    c, _ = gsc.render(cam, w, h, spp, max_depth=5, seed=1)
    assert c.tobytes() != a.tobytes()
    rgb = pbrt_hip.film_to_rgb(a)
    assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and 0.3 < rgb.mean() < 1.0   # rho 0.5 cloud under L = 1


def test_config4_256spp_rank_shares(hip_ctx, config3):
    """BASELINE config 4 at its stated size: the 1 M-triangle scene at 1920x1080x256 spp. The four rank shares
    (tile_world = 4: what each of four GPUs renders before the RCCL film reduce) sum to the one-GPU frame, and a crop of
    that frame is the oracle's at the same 256 spp."""
    sc, gsc = config3
    w, h, spp = 1920, 1080, 256
    cam = scenes.random_triangles_camera(w, h)
    full, st = gsc.render(cam, w, h, spp, max_depth=5, seed=0)
    assert st["camera_samples"] == w * h * spp
    parts, rays = [], 0
    for r in range(4):
        f, st_r = gsc.render(cam, w, h, spp, max_depth=5, seed=0, tile_rank=r, tile_world=4)
        parts.append(f)
        rays += st_r["rays_closest"] + st_r["rays_shadow"]
    assert rays == st["rays_closest"] + st["rays_shadow"]     # the shares trace exactly the frame's rays
    owned = sum((p[..., 3] >= spp).astype(np.int32) for p in parts)
    assert np.all(owned == 1)
    total = sum(parts)
    plain = full[..., 3] == spp
    assert plain.mean() > 0.95
    assert np.array_equal(total[plain], full[plain])
    assert np.allclose(total, full, rtol=1e-6, atol=1e-5)
    bounds = (928, 508, 992, 572)
    osc = oracle.OracleScene(sc)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, spp, max_depth=5, seed=0, bounds=bounds, n_threads=16)
    osc.close()
    crop = (slice(bounds[1], bounds[3]), slice(bounds[0], bounds[2]))
    film_g, st_g = gsc.render(cam, w, h, spp, max_depth=5, seed=0, bounds=bounds)
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g[crop]), oracle.film_to_rgb(film_c[crop])
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    inner = (slice(bounds[1] + 1, bounds[3] - 1), slice(bounds[0] + 1, bounds[2] - 1))
    assert np.array_equal(full[inner][..., 3], film_g[inner][..., 3])
    assert np.allclose(full[inner], film_g[inner], rtol=1e-6, atol=1e-5)
    rmse = float(np.sqrt(np.mean((rgb_g.astype(np.float64) - rgb_c) ** 2)))
    assert rmse <= 1e-6     # north_star: 1e-4


def test_config3_ray_batch_properties(hip_ctx, config3):
    sc, gsc = config3
    rays = scenes.random_rays(4_000_000, 77, origin_extent=1.2)
    hits = gsc.intersect(rays)
    occl = gsc.intersect_p(rays)
    found = hits["prim_id"] >= 0
    assert np.array_equal(found, occl.astype(bool))          # closest-hit and any-hit agree on occlusion
    assert found.mean() > 0.5
    # t_max just beyond the hit keeps it; t_max short of it finds nothing (nothing lies before the closest hit)
    sub = np.nonzero(found)[0][:500_000]
    r2 = rays[sub].copy()
    r2["t_max"] = np.nextafter(hits["t"][sub], np.float32(np.inf))
    h2 = gsc.intersect(r2)
    assert np.array_equal(h2["prim_id"], hits["prim_id"][sub]) and np.array_equal(h2["t"], hits["t"][sub])
    r3 = rays[sub].copy()
    r3["t_max"] = hits["t"][sub] * np.float32(0.999)
    h3 = gsc.intersect(r3)
    assert np.all(h3["prim_id"] < 0)
    # barycentrics reconstruct the hit point on the ray
    tri = sc["positions"][sc["indices"][hits["prim_id"][sub]]]
    p = (hits["b0"][sub, None] * tri[:, 0] + hits["b1"][sub, None] * tri[:, 1] + hits["b2"][sub, None] * tri[:, 2])
    q = rays["o"][sub] + hits["t"][sub, None] * rays["d"][sub]
    assert np.abs(p - q).max() < 1e-4


def test_config5_instanced_4k(hip_ctx):
    """Config 5 geometry at its full resolution (10 000 triangles x 1000 TransformedPrimitives, 3840x2160, depth 16,
    matte / mirror / glass by instance) at 2 spp: a 128x128 crop against the oracle, determinism over the pass
    split, the tile partition of two ranks, and closest-hit / any-hit agreement on a ray batch."""
    w, h = 3840, 2160
    sc = scenes.instanced_scene(10_000, 1000)
    cam = scenes.instanced_camera(w, h)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    kw = dict(max_depth=16, light_strategy=1, seed=5)
    bounds = (1856, 1016, 1984, 1144)
    osc = oracle.OracleScene(sc)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, 2, bounds=bounds, n_threads=16, **kw)
    osc.close()
    film_g, st_g = gsc.render(cam, w, h, 2, bounds=bounds, **kw)
    crop = (slice(bounds[1], bounds[3]), slice(bounds[0], bounds[2]))
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g[crop]), oracle.film_to_rgb(film_c[crop])
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    a, st_a = gsc.render(cam, w, h, 2, **kw)
    b, st_b = gsc.render(cam, w, h, 2, spp_per_pass=1, **kw)
    assert a.tobytes() == b.tobytes() and st_a["rays_closest"] == st_b["rays_closest"]
    assert np.array_equal(a[crop], film_g[crop])            # the crop render is the frame's crop
    parts = [gsc.render(cam, w, h, 2, tile_rank=r, tile_world=2, **kw)[0] for r in range(2)]
    assert np.allclose(parts[0] + parts[1], a, rtol=1e-6, atol=1e-6)
    assert np.all(a[..., 3] >= 2) and np.isfinite(a).all()
    rays = scenes.random_rays(1_000_000, 9, origin_extent=5.0)
    hits, occl = gsc.intersect(rays), gsc.intersect_p(rays)
    assert np.array_equal(hits["prim_id"] >= 0, occl.astype(bool))
    assert np.all((hits["instance_id"] >= 0) == (hits["prim_id"] >= 0))
    gsc.close()


def test_config5_instanced_4k_128spp(hip_ctx):
    """Config 5 at its stated size: 3840x2160 x 128 spp, depth 16 (5.9 G rays, about 6 s on one MI355X). The full
    frame contains the crop render sample for sample; the crop equals the oracle's at a CPU-affordable 8 spp."""
    w, h, spp = 3840, 2160, 128
    sc = scenes.instanced_scene(10_000, 1000)
    cam = scenes.instanced_camera(w, h)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    kw = dict(max_depth=16, light_strategy=1, seed=5)
    a, st_a = gsc.render(cam, w, h, spp, **kw)
    assert st_a["camera_samples"] == w * h * spp
    # a sample whose film position x + u rounds to a whole pixel in f32 also lands on the neighbour (film.rs:263-264):
    # ulp(x) is 2^-12 beyond x = 2048, so with 128 samples a few percent of the pixels receive one extra weight
    assert np.all(a[..., 3] >= spp) and (a[..., 3] != spp).mean() < 0.1 and np.all(a[..., 3] <= spp + 4) and np.isfinite(a).all()
    rgb = pbrt_hip.film_to_rgb(a)
    assert rgb.min() >= 0.0 and 0.05 < rgb.mean() < 2.0
    bounds = (1888, 1048, 1952, 1112)
    crop = (slice(bounds[1], bounds[3]), slice(bounds[0], bounds[2]))
    inner = (slice(bounds[1] + 1, bounds[3] - 1), slice(bounds[0] + 1, bounds[2] - 1))
    g, _ = gsc.render(cam, w, h, spp, bounds=bounds, **kw)
    assert np.array_equal(a[inner][..., 3], g[inner][..., 3])
    assert np.allclose(a[inner], g[inner], rtol=1e-6, atol=1e-5)
    osc = oracle.OracleScene(sc)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, 8, bounds=bounds, n_threads=16, **kw)
    osc.close()
    film_g, st_g = gsc.render(cam, w, h, 8, bounds=bounds, **kw)
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g[crop]), oracle.film_to_rgb(film_c[crop])
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    gsc.close()


@pytest.mark.parametrize("sampler", [("stratified", 4, 4, True, 4), ("zerotwo", 4), ("halton",)])
def test_config3_samplers_full_frame(hip_ctx, config3, sampler):
    """The tabulating / Halton samplers at config 3's size (1 M triangles, 1920x1080, 16 spp): a 128x128 crop
    against the oracle, and the full frame independent of the pass split (per-pixel tables, 1.6 GB here)."""
    sc, gsc = config3
    w, h = 1920, 1080
    cam = scenes.random_triangles_camera(w, h)
    bounds = (896, 476, 1024, 604)
    osc = oracle.OracleScene(sc)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, 16, max_depth=5, seed=2, bounds=bounds,
                              n_threads=16, sampler=sampler)
    osc.close()
    film_g, st_g = gsc.render(cam, w, h, 16, max_depth=5, seed=2, bounds=bounds, sampler=sampler)
    crop = (slice(bounds[1], bounds[3]), slice(bounds[0], bounds[2]))
    assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film_g[crop]), oracle.film_to_rgb(film_c[crop])
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))
    a, st_a = gsc.render(cam, w, h, 16, max_depth=5, seed=2, sampler=sampler)
    b, st_b = gsc.render(cam, w, h, 16, max_depth=5, seed=2, sampler=sampler, spp_per_pass=5)
    assert a.tobytes() == b.tobytes() and st_a["rays_closest"] == st_b["rays_closest"]
    # inside the crop the frame is the crop render (its border also receives the samples that neighbours outside the
    # crop place exactly on a pixel corner: Halton's first sample of a pixel has offset 0)
    inner = (slice(bounds[1] + 1, bounds[3] - 1), slice(bounds[0] + 1, bounds[2] - 1))
    assert np.array_equal(a[inner][..., 3], film_g[inner][..., 3])
    assert np.allclose(a[inner], film_g[inner], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("w,h,spp", [(1920, 1080, 64), (3840, 2160, 8)])
def test_film_properties_at_full_size(hip_ctx, w, h, spp):
    """The film stage at BASELINE's frame sizes, through properties that need no oracle (tests/closed_forms_film.py at small sizes):
    a constant-radiance scene reconstructs to that constant in EVERY pixel under the 0.5 box and under a Gaussian of radius 2
    (k_film_accumulate / k_film_splat over 133 M / 66 M samples, several passes), the box film's weight channel holds every sample
    once (bench.py's film_check), and the frame is the sum of eight ranks' Morton-dealt tile shares."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import closed_forms_film as cf
    le = (0.7, 1.3, 2.1)
    g = pbrt_hip.Scene(hip_ctx, cf.sky_scene(le))
    cam = cf.sky_camera(w, h)
    through = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]]) @ \
        np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]]) @ np.array(le)
    box, st = g.render(cam, w, h, spp, max_depth=5, seed=1)
    assert st["camera_samples"] == w * h * spp == st["rays_closest"] and st["rays_shadow"] == 0
    fc = bench.check_film_weights(box[..., 3].astype(np.float64), spp)
    assert fc["ok"] and fc["excess"] > 0, fc          # (samples that round onto a pixel border count in both neighbours)
    assert np.all(np.abs(pbrt_hip.film_to_rgb(box) - through) <= 2e-6 * max(le))
    clamped, _ = g.render(cam, w, h, spp, max_depth=5, seed=1, max_sample_luminance=1.0)   # film.rs:253-255 through the same kernels
    assert np.all(np.abs(cf.luminance(pbrt_hip.film_to_rgb(clamped)) - 1.0) <= 3e-6)
    shares = sum(g.render(cam, w, h, spp, max_depth=5, seed=1, tile_rank=r, tile_world=8)[0] for r in range(8))
    # bit for bit wherever a pixel's samples all come from its own tile; a pixel on a tile border may also hold border samples of
    # the neighbouring tiles (other ranks under the Morton deal), and then only the ORDER of the float additions differs
    ys, xs = np.mgrid[0:h, 0:w]
    inner = ((xs % 16 != 0) & (xs % 16 != 15) & (ys % 16 != 0) & (ys % 16 != 15))
    assert np.array_equal(shares[inner], box[inner])
    assert np.all(np.abs(shares - box) <= 2.5e-7 * np.abs(box)) and np.array_equal(shares[..., 3], box[..., 3])
    wide, _ = g.render(cam, w, h, spp, max_depth=5, seed=1, filter=pbrt_hip.filter_table("gaussian", 2.0, 2.0, 2.0))
    assert wide[..., 3].min() > 0.5 * spp
    # ~1600 float atomics per pixel in arbitrary order (64 spp x a 5 x 5 footprint), then xyz_to_rgb's cancelling sums of magnitude
    # 3 |le|: 2e-5 of the largest channel (the small-size closed form with its ordered sums holds 2e-6; north_star's budget is 1e-4)
    err = np.abs(pbrt_hip.film_to_rgb(wide) - through)
    assert err.max() <= 2e-5 * max(le) and float(np.sqrt(np.mean(err ** 2))) <= 3e-6 * max(le), (err.max(), np.sqrt(np.mean(err ** 2)))
    g.close()
