"""pbrt_hip_context_set_traversal: the three traversal kernels compute BVHAccel::intersect / intersect_p
(src/accelerators/bvh.rs:828-932) bit for bit — the 4-wide records (default), the binary records with a stack, and the
binary records walked without one (parent links + a 64-bit trail, csrc/trace_stackless.h: the kernel BASELINE.json's
north_star names). Every mode against the oracle, and against each other on the film."""
import contextlib

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu

MODES = [pbrt_hip.TRAVERSAL_AUTO, pbrt_hip.TRAVERSAL_STACK, pbrt_hip.TRAVERSAL_STACKLESS]


@contextlib.contextmanager
def traversal(ctx, mode):
    ctx.set_traversal(mode)
    try:
        yield
    finally:
        ctx.set_traversal(pbrt_hip.TRAVERSAL_AUTO)      # the context is shared by the session


def _bits_equal(gpu, cpu, what):
    for f in ("prim_id", "t", "b0", "b1", "b2"):
        a, b = gpu[f], cpu[f]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        bad = np.flatnonzero(a != b)
        assert len(bad) == 0, (what, f, bad[:5], gpu[bad[:3]], cpu[bad[:3]])


def _aimed_rays(verts, idx, seed=7):
    """Rays through every vertex, edge midpoint and centroid, a fifth of them with one direction component zeroed
    (tests/test_gpu_intersect.py::test_t_max_that_moves_up_by_an_ulp)."""
    tri = verts[idx]
    targets = np.concatenate([tri.reshape(-1, 3), (tri[:, 0] + tri[:, 1]) * np.float32(0.5), tri.mean(axis=1)])
    n = 3 * len(targets)
    rays = scenes.random_rays(n, seed, origin_extent=2.0)
    tgt = targets[np.arange(n) % len(targets)]
    scale = np.maximum(np.abs(tgt).max(axis=1, keepdims=True), 1.0).astype(np.float32)
    rays["o"] = (tgt + rays["o"] * scale).astype(np.float32)
    rays["d"] = (tgt - rays["o"]).astype(np.float32)
    k = np.arange(n)
    par = k % 5 == 0
    rays["d"][par, k[par] % 3] = 0.0
    rays["d"][np.all(rays["d"] == 0, axis=1)] = (0.0, 0.0, 1.0)
    return np.ascontiguousarray(rays)


def _plain(verts, idx):
    return dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
                materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))


def _fan_mesh():
    g = np.array([[x, y, 0.0] for y in range(7) for x in range(7)], dtype=np.float32)
    g[:, 2] = (np.sin(g[:, 0] * 1.3) + np.cos(g[:, 1] * 0.7)).astype(np.float32)
    verts = (g * np.float32(333.0) + np.float32(-1000.0)).astype(np.float32)
    quad = [(y * 7 + x, y * 7 + x + 1, (y + 1) * 7 + x + 1, (y + 1) * 7 + x) for y in range(6) for x in range(6)]
    return verts, np.array([t for a, b, c, d in quad for t in ((a, b, c), (a, c, d))], dtype=np.int32)


def _chain(n):
    """SPLIT_MIDDLE peels one triangle off per level: a tree n levels deep (test_too_deep_tree_is_refused)."""
    x = (3.0 ** np.arange(n)).astype(np.float32)
    pos = np.zeros((3 * n, 3), dtype=np.float32)
    pos[0::3, 0], pos[1::3, 0], pos[2::3, 0] = x, x, x
    pos[1::3, 1], pos[2::3, 2] = 1e-3, 1e-3
    return pos, np.arange(3 * n, dtype=np.int32).reshape(n, 3)


CASES = {
    "cornell": lambda: (scenes.cornell_box(), 4, pbrt_hip.SPLIT_SAH, None),
    "rand20k": lambda: (scenes.random_triangles(20_000, seq=3, size=0.05), 4, pbrt_hip.SPLIT_SAH, None),
    "rand20k leaves of one": lambda: (scenes.random_triangles(20_000, seq=4, size=0.05), 1, pbrt_hip.SPLIT_EQUAL_COUNTS, None),
    "mixed hlbvh": lambda: (scenes.mixed_materials_scene(), 4, pbrt_hip.SPLIT_HLBVH, None),
    "one triangle": lambda: (_plain(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32),
                                    np.array([[0, 1, 2]], dtype=np.int32)), 4, pbrt_hip.SPLIT_SAH, None),
    "fan mesh (t_max moves up)": lambda: (_plain(*_fan_mesh()), 1, pbrt_hip.SPLIT_SAH, _aimed_rays(*_fan_mesh())),
    "chain 64 levels deep": lambda: (_plain(*_chain(64)), 1, pbrt_hip.SPLIT_MIDDLE, None),
}


@pytest.mark.parametrize("name", list(CASES))
def test_every_traversal_is_the_reference_walk(hip_ctx, name):
    sc, max_prims, split, rays = CASES[name]()
    osc = oracle.OracleScene(sc, max_prims, split)
    gsc = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=max_prims, split_method=split)
    if rays is None:
        n = 60_000
        rays = scenes.random_rays(n, 11, origin_extent=600.0 if name == "cornell" else 1.5)
        if name == "cornell":
            rays["o"] = np.abs(rays["o"]) * np.float32(0.9)
        if name.startswith("chain"):
            # along the chain (every level's box is entered), from both ends, and across it
            rays["o"][:, 0] = np.where(np.arange(n) % 2 == 0, -1.0, 4e30).astype(np.float32)
            rays["o"][:, 1:] *= np.float32(1e-3)
            rays["d"] = np.where((np.arange(n) % 2 == 0)[:, None], [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]).astype(np.float32)
            rays["d"][:, 1:] += (rays["o"][:, 1:] * np.float32(-1e-3)).astype(np.float32)
        rays["t_max"][::4] = np.float32(300.0 if name == "cornell" else 0.75)
        if name.startswith("chain"):
            rays["t_max"][:] = np.inf
    cpu, _ = osc.intersect(rays)
    cpu_p, _ = osc.intersect_p(rays)
    assert (cpu["prim_id"] >= 0).any()
    for mode in MODES:
        with traversal(hip_ctx, mode):
            _bits_equal(gsc.intersect(rays), cpu, (name, mode))
            assert np.array_equal(gsc.intersect_p(rays), cpu_p), (name, mode)
    gsc.close()
    osc.close()


def test_stackless_on_a_device_built_tree(hip_ctx):
    """The parent links of a tree built and laid out on the GPU (hlbvh_gpu.hip k_parent_links)."""
    sc = scenes.random_triangles(50_000, seq=8, size=0.03)
    osc = oracle.OracleScene(sc, split_method=pbrt_hip.SPLIT_HLBVH)
    gsc = pbrt_hip.Scene(hip_ctx, sc, device_build=True)
    rays = scenes.random_rays(80_000, 5, origin_extent=1.5)
    cpu, _ = osc.intersect(rays)
    for mode in MODES:
        with traversal(hip_ctx, mode):
            _bits_equal(gsc.intersect(rays), cpu, mode)
    gsc.close()
    osc.close()


@pytest.mark.parametrize("integrator,kw", [(pbrt_hip.INTEGRATOR_PATH, dict(max_depth=6, light_strategy=1)),
                                           (pbrt_hip.INTEGRATOR_DIRECT, dict(max_depth=3, light_strategy=0)),
                                           (pbrt_hip.INTEGRATOR_AO, dict(ao_samples=4, cos_sample=True))])
def test_the_film_does_not_depend_on_the_traversal(hip_ctx, integrator, kw):
    """The wavefront renderer under each kernel: the same bits on the film, the same ray counts."""
    w, h = 96, 64
    sc, cam = scenes.mixed_materials_scene(), scenes.random_triangles_camera(w, h)
    g = pbrt_hip.Scene(hip_ctx, sc)
    out = []
    for mode in MODES:
        with traversal(hip_ctx, mode):
            out.append(g.render(cam, w, h, 4, integrator=integrator, seed=17, **kw))
    g.close()
    assert out[0][0].tobytes() == out[1][0].tobytes() == out[2][0].tobytes()
    assert len({(st["rays_closest"], st["rays_shadow"]) for _, st in out}) == 1
    assert out[0][1]["rays_closest"] > w * h


def test_stackless_refuses_what_it_does_not_cover(hip_ctx):
    with pytest.raises(pbrt_hip.PbrtHipError, match="PBRT_TRAVERSAL"):
        hip_ctx.set_traversal(3)
    rays = scenes.random_rays(100, 1)
    inst = pbrt_hip.Scene(hip_ctx, scenes.instanced_scene(200, 5, extent=1.5))
    sph = pbrt_hip.Scene(hip_ctx, dict(scenes.cornell_box(), spheres=np.array([[250, 250, 250, 60, 0, -1, 0, 0]], dtype=np.float32)))
    plain = pbrt_hip.Scene(hip_ctx, scenes.cornell_box())
    with traversal(hip_ctx, pbrt_hip.TRAVERSAL_STACKLESS):
        for g in (inst, sph):
            with pytest.raises(pbrt_hip.PbrtHipError, match="single-level triangle scenes only"):
                g.intersect(rays)
            with pytest.raises(pbrt_hip.PbrtHipError, match="single-level triangle scenes only"):
                g.intersect_p(rays)
        with pytest.raises(pbrt_hip.PbrtHipError, match="single-level triangle scenes only"):
            inst.render(scenes.instanced_camera(16, 16, 1.5), 16, 16, 1, max_depth=2)
        hip_ctx.set_counting(1)
        try:
            with pytest.raises(pbrt_hip.PbrtHipError, match="no counting variant"):
                plain.intersect(rays)
        finally:
            hip_ctx.set_counting(0)
        assert len(plain.intersect(rays)) == 100
    for g in (inst, sph, plain):
        assert len(g.intersect(rays)) == 100          # back on TRAVERSAL_AUTO
        g.close()


def _degenerate_above_matte(form):
    """A matte triangle T of size 1e-9 through the origin and, 1e-12 above it, a triangle D with 5e-12 legs: D's geometric
    normal (p2 - p0) x (p1 - p0) has squared length 6e-46, which is 0 in f32 — Triangle::intersect rejects it
    (triangle.rs:212-216) while Triangle::intersect_p (= intersect_test alone, triangle.rs:318-321) accepts it. From T's hit
    point D covers 43 % of a cosine-sampled hemisphere. Constant environment light. `form`: flat / instanced (one identity
    instance, INST = 1) / two-level with D in world space beside an instance holding T / the other way round (INST = 2)."""
    f = np.float32
    s, d, h = f(1e-9), f(5e-12), f(1e-12)
    T = np.array([[-s, -s, 0], [s, -s, 0], [0, s, 0]], dtype=np.float32)
    D = np.array([[-d / 2, -d / 2, h], [d / 2, -d / 2, h], [-d / 2, d / 2, h]], dtype=np.float32)
    tri = np.array([[0, 1, 2]], dtype=np.int32)
    mats = scenes._materials([(scenes.MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)])
    env = scenes._lights([(scenes.LIGHT_INFINITE, (1.0, 1.0, 1.0), -1, 0, 1)])
    ident = np.zeros((1, 2, 4, 4), dtype=np.float32)
    ident[0, 0] = ident[0, 1] = np.eye(4)
    both = dict(positions=np.concatenate([T, D]), indices=np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int32),
                tri_material=np.zeros(2, dtype=np.int32), tri_light=np.full(2, -1, dtype=np.int32))
    if form == "flat":
        return dict(both, materials=mats, lights=env)
    if form == "instanced":
        return dict(both, materials=mats, lights=env, instances=ident, instance_material=np.array([-1], dtype=np.int32))
    inner, outer = (T, D) if form == "two-level, D in world space" else (D, T)
    return dict(objects=[dict(positions=inner, indices=tri, tri_material=np.zeros(1, dtype=np.int32))], instances=ident,
                instance_object=np.zeros(1, dtype=np.int32), instance_material=np.array([-1], dtype=np.int32),
                world=dict(positions=outer, indices=tri, tri_material=np.zeros(1, dtype=np.int32), tri_light=np.full(1, -1, dtype=np.int32)),
                materials=mats, lights=env)


@pytest.mark.parametrize("form", ["flat", "instanced", "two-level, D in world space", "two-level, D inside the instance"])
def test_boolean_mis_rays_skip_what_triangle_intersect_rejects(hip_ctx, form):
    """the `strict` branches of RS_MIS_BOOL. estimate_direct's BSDF-sampled ray goes through
    Scene::intersect (integrator.rs:232-262), so a triangle Triangle::intersect rejects must not make `found` true although the
    any-hit walk that traces such a ray would stop at it. Here 43 % of the MIS rays can only hit such a triangle: a kernel that
    lost a `strict` check would turn their environment contribution into 0. Every kernel (wide / stack / stackless) gives the
    oracle's radiance, and the same numbers as the counted run in which the MIS rays are walked to their closest hit."""
    sc = _degenerate_above_matte(form)
    n = 2048
    rays = np.zeros(n, dtype=scenes.RAY_DTYPE)
    rays["o"], rays["d"], rays["t_max"] = (0.0, 0.0, 1e-6), (0.0, 0.0, -1.0), np.inf
    keys = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    osc = oracle.OracleScene(sc)
    cpu, st_c = osc.li(rays, keys, max_depth=1)
    # the test has power: from just above T a large share of cosine-distributed directions hit D under intersect_p and nothing
    # under intersect
    u = scenes.pcg32_float(77, 2 * n).reshape(n, 2).astype(np.float64)
    r, phi = np.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    probe = np.zeros(n, dtype=scenes.RAY_DTYPE)
    probe["o"], probe["t_max"] = (0.0, 0.0, 1e-16), np.inf
    probe["d"] = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u[:, 0])], 1).astype(np.float32)
    any_c, closest_c = osc.intersect_p(probe)[0], osc.intersect(probe)[0]
    assert 0.3 < any_c.mean() < 0.6 and (closest_c["prim_id"] < 0).all()
    osc.close()
    g = pbrt_hip.Scene(hip_ctx, sc)
    modes = [pbrt_hip.TRAVERSAL_AUTO, pbrt_hip.TRAVERSAL_STACK] + ([pbrt_hip.TRAVERSAL_STACKLESS] if form == "flat" else [])
    for mode in modes:
        with traversal(hip_ctx, mode):
            assert np.array_equal(g.intersect_p(probe), any_c) and (g.intersect(probe)["prim_id"] < 0).all()
            gpu, st = g.li(rays, keys, max_depth=1)
            assert st["rays_closest"] + st["rays_shadow"] == st_c["rays"]
            assert np.all(np.abs(gpu - cpu) <= 1e-5 * np.maximum(1.0, np.abs(cpu))), (mode, np.abs(gpu - cpu).max())
            if mode != pbrt_hip.TRAVERSAL_STACKLESS:   # (the stackless kernel has no counting variant)
                hip_ctx.set_counting(1)
                try:
                    counted, st_n = g.li(rays, keys, max_depth=1)
                finally:
                    hip_ctx.set_counting(0)
                assert counted.tobytes() == gpu.tobytes() and (st_n["rays_closest"], st_n["rays_shadow"]) == (st["rays_closest"], st["rays_shadow"])
    # what a lost `strict` would give: every MIS ray that D stops contributes nothing -> the mean radiance drops by a third or more
    assert cpu.mean() > 0.4
    g.close()
