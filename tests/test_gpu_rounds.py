"""PBRT_TRAVERSAL_ROUNDS (csrc/trace_rounds.h): TransformedPrimitive::intersect (src/core/primitive.rs:136-159) with the top-level
walk and the walks inside the instances in launches of their own. A ray still meets its instances one after the other, in the
reference's order, with the t_max the last one left — so films, ray counts and li batches are the fused kernel's bit for bit and
the oracle's within the render tolerance."""
import contextlib

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def traversal(ctx, mode):
    ctx.set_traversal(mode)
    try:
        yield
    finally:
        ctx.set_traversal(pbrt_hip.TRAVERSAL_AUTO)      # the context is shared by the session


def _both(ctx, g, cam, w, h, spp, **kw):
    fused, st_f = g.render(cam, w, h, spp, **kw)
    with traversal(ctx, pbrt_hip.TRAVERSAL_ROUNDS):
        rounds, st_r = g.render(cam, w, h, spp, **kw)
    assert rounds.tobytes() == fused.tobytes()
    assert (st_r["rays_closest"], st_r["rays_shadow"]) == (st_f["rays_closest"], st_f["rays_shadow"])
    return rounds, st_r


@pytest.mark.parametrize("integrator,max_depth", [(0, 16), (1, 4), (3, 1)])
def test_rounds_equal_the_fused_kernel_and_the_oracle(hip_ctx, integrator, max_depth):
    """Matte / mirror / glass instances (config 5's shape, small): path, direct lighting (specular recursion: closest-hit rays that
    start INSIDE instances) and ambient occlusion (any-hit rays only)."""
    w, h, spp = 96, 64, 4
    sc = scenes.instanced_scene(2500, 60, extent=1.5)
    cam = scenes.instanced_camera(w, h, 1.5)
    g = pbrt_hip.Scene(hip_ctx, sc)
    kw = dict(integrator=integrator, max_depth=max_depth, seed=41, light_strategy=1 if integrator == 0 else 0)
    if integrator == 3:
        kw = dict(integrator=integrator, ao_samples=8, cos_sample=True, seed=41)
    film, st = _both(hip_ctx, g, cam, w, h, spp, **kw)
    g.close()
    osc = oracle.OracleScene(sc)
    film_c, st_c = osc.render(scenes.camera_dict_to_floats(cam), w, h, spp, **kw)
    osc.close()
    assert st["rays_closest"] + st["rays_shadow"] == st_c["rays"]
    rgb_g, rgb_c = pbrt_hip.film_to_rgb(film), oracle.film_to_rgb(film_c)
    assert np.all(np.abs(rgb_g - rgb_c) <= 1e-5 * np.maximum(1.0, np.abs(rgb_c)))


def test_rounds_many_entries_per_ray_and_queue_slices(hip_ctx):
    """Overlapping instances (a ray enters many of them: many rounds), instances without a material override, leaves of several
    instances in the top-level tree, and a frame traced in several passes."""
    w, h, spp = 160, 96, 6
    sc = scenes.instanced_scene(400, 300, extent=0.9, base_extent=0.5, tri_size=0.08)
    n = len(sc["indices"])
    sc["tri_material"] = (np.arange(n) % 3).astype(np.int32)
    sc["instance_material"] = np.where(np.arange(300) % 2 == 0, -1, np.arange(300) % 3).astype(np.int32)
    cam = scenes.instanced_camera(w, h, 0.9)
    for tlas_max in (1, 4):
        g = pbrt_hip.Scene(hip_ctx, sc, bvh=pbrt_hip.build_two_level(sc, tlas_max_prims=tlas_max))
        _both(hip_ctx, g, cam, w, h, spp, max_depth=12, seed=7)
        _both(hip_ctx, g, cam, w, h, spp, max_depth=12, seed=7, spp_per_pass=1)
        g.close()


def test_rounds_li_batch_and_other_scenes(hip_ctx):
    """pbrt_hip_li under the same switch; scenes the rounds do not cover (one level, the general two-level form) run as AUTO."""
    sc = scenes.instanced_scene(1500, 30, extent=1.2)
    g = pbrt_hip.Scene(hip_ctx, sc)
    rays = scenes.random_rays(20_000, 3, origin_extent=2.0)
    keys = np.arange(len(rays), dtype=np.uint64) * np.uint64(7919)
    a, _ = g.li(rays, keys, max_depth=8)
    with traversal(hip_ctx, pbrt_hip.TRAVERSAL_ROUNDS):
        b, _ = g.li(rays, keys, max_depth=8)
        hits = g.intersect(rays)          # batch calls: the fused kernel
    assert a.tobytes() == b.tobytes()
    assert hits.tobytes() == g.intersect(rays).tobytes()
    g.close()
    w, h = 64, 48
    for sc2, cam in ((scenes.cornell_box(), scenes.cornell_camera(w, h)), (scenes.two_level_scene(30), scenes.two_level_camera(w, h))):
        g2 = pbrt_hip.Scene(hip_ctx, sc2)
        _both(hip_ctx, g2, cam, w, h, 4, max_depth=5, seed=2)
        g2.close()
