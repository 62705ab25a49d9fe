"""GPU parity of the batch Integrator::li entry point (pbrt_hip_li; src/core/integrator.rs:29-42, :452).

 * against the oracle's li on the same rays and stream keys: same ray count (every random decision equal), radiance
   within 1e-5 * max(1, |L|) per channel (the per-path sums are evaluated in the same order; the tolerance is the render
   tests' per-pixel bound);
 * against pbrt_hip_render: fed pbrt_hip_camera_rays' rays and keys (5 values drawn before li, the CameraSample),
   li returns render's samples: accumulated per pixel in sample order on the host they give render's film BIT FOR BIT.
"""
import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


def _rays_into(sc_name, n, seq):
    rays = scenes.random_rays(n, seq, origin_extent=1.2)   # the random-triangle scenes fill [-1, 1]^3
    if sc_name == "cornell":
        rays["o"] = np.abs(rays["o"]) * np.float32(200.0) + np.float32(40.0)   # inside the box
    return rays


@pytest.mark.parametrize("name,integrator,kw", [
    ("cornell", pbrt_hip.INTEGRATOR_PATH, dict(max_depth=8, light_strategy=1)),
    ("mixed", pbrt_hip.INTEGRATOR_PATH, dict(max_depth=16, light_strategy=0)),
    ("rand20k", pbrt_hip.INTEGRATOR_PATH, dict(max_depth=5)),
    ("cornell", pbrt_hip.INTEGRATOR_DIRECT, dict(max_depth=3, light_strategy=0)),
    ("mixed", pbrt_hip.INTEGRATOR_WHITTED, dict(max_depth=5)),
    ("rand20k", pbrt_hip.INTEGRATOR_AO, dict(ao_samples=8)),
])
def test_li_equals_oracle_li(hip_ctx, name, integrator, kw):
    sc = {"cornell": scenes.cornell_box, "mixed": scenes.mixed_materials_scene,
          "rand20k": lambda: scenes.random_triangles(20_000, seq=3, size=0.05)}[name]()
    n = 3001   # ragged: not a multiple of the 256-path tiles
    rays = _rays_into(name, n, 17)
    keys = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    for skip in (0, 5):
        cpu, st_c = osc.li(rays, keys, integrator=integrator, draws_before_li=skip, **kw)
        gpu, st_g = gsc.li(rays, keys, integrator=integrator, draws_before_li=skip, **kw)
        assert st_g["rays_closest"] + st_g["rays_shadow"] == st_c["rays"]
        assert np.all(np.abs(gpu - cpu) <= 1e-5 * np.maximum(1.0, np.abs(cpu))), np.abs(gpu - cpu).max()
    assert cpu.max() > 0.01
    # skipping draws changes the stream: the two runs must differ somewhere
    assert not np.array_equal(cpu, osc.li(rays, keys, integrator=integrator, draws_before_li=0, **kw)[0])
    # an empty batch is a no-op
    e, st = gsc.li(rays[:0], keys[:0])
    assert e.shape == (0, 3) and st["rays_closest"] == 0
    gsc.close()
    osc.close()


def test_li_on_camera_rays_reproduces_render_bit_for_bit(hip_ctx):
    w, h, spp = 80, 48, 6
    sc = scenes.cornell_box()
    cam = scenes.cornell_camera(w, h)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    film, st = gsc.render(cam, w, h, spp, max_depth=8, seed=9)
    rays, keys, pfilm, pix = gsc.camera_rays(cam, w, h, spp, seed=9)
    valid = pix[:, 0] >= 0
    assert valid.sum() == w * h * spp and np.all(rays["t_max"][~valid] < 0)
    # keys are the documented (pixel, sample) streams: seed ^ ((y * W + x) * spp + s)
    x, y, s = (pix[valid, k].astype(np.int64) for k in range(3))
    assert np.array_equal(keys[valid], np.uint64(9) ^ ((y * w + x) * spp + s).astype(np.uint64))
    rgb, st_li = gsc.li(rays[valid], keys[valid], max_depth=8, draws_before_li=5)
    assert st_li["rays_closest"] + st_li["rays_shadow"] == st["rays_closest"] + st["rays_shadow"]
    # FilmTile::add_sample with the 0.5 box + merge_film_tile, samples of a pixel in sample order (film.rs:252-295, 111-123)
    f32 = np.float32
    acc = np.zeros((h, w, 3), dtype=np.float32)
    order = np.lexsort((s, x, y))
    # the film position decides the pixel (ceil / floor in add_sample): here it is the generating pixel for every sample
    assert np.array_equal(np.floor(pfilm[valid]).astype(np.int64), np.stack([x, y], axis=1))
    L = rgb.copy()
    lum = f32(0.212671) * L[:, 0] + f32(0.715160) * L[:, 1] + f32(0.072169) * L[:, 2]
    bad = np.isnan(L).any(axis=1) | (lum < f32(-1e-5)) | np.isinf(lum)      # integrator.rs:455
    L[bad] = 0
    for k in range(spp):            # k-th sample of every pixel at once: per pixel the additions stay in sample order
        sel = order[k::spp]
        assert np.array_equal(s[sel], np.full(len(sel), k))
        acc[y[sel], x[sel]] += L[sel]
    xyz = np.stack([f32(0.412453) * acc[..., 0] + f32(0.357580) * acc[..., 1] + f32(0.180423) * acc[..., 2],
                    f32(0.212671) * acc[..., 0] + f32(0.715160) * acc[..., 1] + f32(0.072169) * acc[..., 2],
                    f32(0.019334) * acc[..., 0] + f32(0.119193) * acc[..., 1] + f32(0.950227) * acc[..., 2]], axis=-1)
    assert np.array_equal(film[..., 3], np.full((h, w), spp, dtype=np.float32))
    assert xyz.astype(np.float32).tobytes() == np.ascontiguousarray(film[..., :3]).tobytes()
    gsc.close()


def test_li_device_buffers_and_tile_share(hip_ctx):
    """camera_rays of one rank's tile share (tile_world = 3) + li == that rank's part of the frame's samples."""
    import ctypes
    w, h, spp = 64, 48, 2
    sc = scenes.random_triangles(5_000, seq=4, size=0.08)
    cam = scenes.random_triangles_camera(w, h)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    full = {}
    rays, keys, _, pix = gsc.camera_rays(cam, w, h, spp, seed=2)
    v = pix[:, 0] >= 0
    rgb, _ = gsc.li(rays[v], keys[v], draws_before_li=5)
    for (x, y, s), c in zip(map(tuple, pix[v]), rgb):
        full[(x, y, s)] = c
    seen = 0
    for rank in range(3):
        r, k, _, p = gsc.camera_rays(cam, w, h, spp, seed=2, tile_rank=rank, tile_world=3)
        vv = p[:, 0] >= 0
        out, _ = gsc.li(r[vv], k[vv], draws_before_li=5)
        for (x, y, s), c in zip(map(tuple, p[vv]), out):
            assert np.array_equal(full[(x, y, s)], c)
        seen += int(vv.sum())
    assert seen == w * h * spp
    gsc.close()
