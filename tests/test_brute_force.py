"""BVHAccel::intersect / intersect_p (bvh.rs:828-933) against the definition that needs no tree: the closest hit is the smallest t
over ALL triangles put to Triangle::intersect's test (itself held to exact rationals in test_exact_rational_pin.py). The brute-force
loop shares nothing with the builders or the walks — the oracle's (o_bvh.h) and, on the GPU, the library's host builder, wide
records and three traversal kernels.
What the reference's algorithm promises, and what is asserted: (1) a ray hits something iff the brute force does (and intersect_p
says so); (2) the hit an aggregate returns IS a hit of the triangle it names: that triangle alone is hit at the same t, bit for
bit; (3) nothing is closer by more than a few ulps. Not "bit for bit the smallest t": once a hit has lowered ray.t_max, a
neighbour sharing the hit edge or vertex whose own t comes out ONE ULP smaller can fail Bounds3f::intersect_p's `t_min < ray.t_max`
(geometry.rs:748: t_min is not widened) or the triangle test's `t_scaled < t_max * det` comparison (triangle.rs:137-141, both
sides rounded) — pbrt-v3 behaves the same; on a cloud of separate triangles — no shared edges or vertices — (4) the two agree bit for bit."""
import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes


def _height_field(n, seq):
    """(n + 1)^2 shared vertices, 2 n^2 triangles: every edge and vertex shared — where ties at equal t live."""
    u = scenes.pcg32_float(seq, (n + 1) * (n + 1)).reshape(n + 1, n + 1)
    xs = np.linspace(-1.0, 1.0, n + 1, dtype=np.float32)
    pos = np.stack([np.repeat(xs[None, :], n + 1, 0), 0.3 * u.astype(np.float32), np.repeat(xs[:, None], n + 1, 1)], axis=-1).reshape(-1, 3)
    idx = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            idx += [[a, a + 1, a + n + 2], [a, a + n + 2, a + n + 1]]
    return pos.astype(np.float32), np.asarray(idx, dtype=np.int32)


def _rays_at(pos, idx, n, seq):
    """Rays from above through random points of the mesh's triangles, through its VERTICES and through its EDGE midpoints."""
    u = scenes.pcg32_float(seq, n * 5).reshape(n, 5)
    tri = pos[idx[(u[:, 0] * len(idx)).astype(int) % len(idx)]]
    kind = np.arange(n) % 3
    b = np.where(kind[:, None] == 0, np.stack([u[:, 1] * (1 - u[:, 2]), u[:, 2], 1 - u[:, 1] * (1 - u[:, 2]) - u[:, 2]], 1),
                 np.where(kind[:, None] == 1, np.array([1.0, 0.0, 0.0]), np.array([0.5, 0.5, 0.0])))
    target = (b[:, :, None] * tri).sum(axis=1).astype(np.float32)
    rays = np.zeros(n, dtype=pbrt_hip.RAY_DTYPE)
    rays["o"] = np.stack([(u[:, 3] * 2 - 1) * 0.5, np.full(n, 2.0), (u[:, 4] * 2 - 1) * 0.5], axis=1).astype(np.float32)
    rays["d"] = target - rays["o"]
    rays["t_max"] = np.inf
    return rays


def _scenes():
    out = {}
    sc = scenes.random_triangles(3000, seq=3, extent=1.0, size=0.12)
    out["cloud"] = (sc["positions"], sc["indices"], scenes.random_rays(3000, 11))
    pos, idx = _height_field(24, 5)
    out["height field"] = (pos, idx, _rays_at(pos, idx, 3000, 6))
    sc = scenes.cornell_box()
    r = scenes.random_rays(2000, 4, origin_extent=500.0)
    r["o"] = np.abs(r["o"])
    out["cornell"] = (sc["positions"], sc["indices"], r)
    return out


def _as_scene(pos, idx):
    return dict(positions=pos, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
                materials=scenes._materials([(scenes.MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([(scenes.LIGHT_INFINITE, (1.0, 1.0, 1.0), -1, 0, 1)]))


def _check_once(name, hits, any_hit, pos, idx, rays, to_object, exact):
    t, prim, inst, ties = oracle.brute_force(pos, idx, rays, to_object)
    found = hits["prim_id"] >= 0
    assert np.array_equal(found, np.isfinite(t)), (name, int((found != np.isfinite(t)).sum()))              # (1)
    assert np.array_equal(any_hit.astype(bool), found), name
    tb, th = t[found].astype(np.float64), hits["t"][found].astype(np.float64)
    assert np.all(th >= tb) and np.all(th - tb <= 4 * 2.0 ** -23 * tb), (name, float(np.max((th - tb) / tb)))  # (3): within 4 ulps
    off = np.flatnonzero(found & ((hits["t"] != t) | (hits["prim_id"] != prim) | ((hits["instance_id"] != inst) if to_object is not None else False)))
    if exact:
        assert not np.any(hits["t"][found] != t[found]), (name, len(off))                                      # (4)
    m = None if to_object is None else np.asarray(to_object).reshape(-1, 16)
    for i in off[:400]:                                                                                        # (2)
        one = idx[hits["prim_id"][i]:hits["prim_id"][i] + 1]
        mi = None if m is None else m[hits["instance_id"][i]:hits["instance_id"][i] + 1]
        assert oracle.brute_force(pos, one, rays[i:i + 1], mi)[0][0] == hits["t"][i], (name, i)
    return t, int(found.sum()), len(off)


def _check(name, intersect, intersect_p, pos, idx, rays, to_object=None, exact=False):
    """Unbounded rays, then the same rays cut short: t_max at 0.5 x, 0.999 x, 1.001 x and 2 x the closest hit — segments that stop
    before the surface, just before it, just behind it and well behind it (shadow rays are such segments: Ray::t_max is what
    intersect_p is about, bvh.rs:881-933). Clear of the one-ulp zone around t_max itself (see the module text)."""
    t, n_hit, n_off = _check_once(name, intersect(rays), intersect_p(rays), pos, idx, rays, to_object, exact)
    cut = rays.copy()
    factor = np.array([0.5, 0.999, 1.001, 2.0], dtype=np.float32)[np.arange(len(rays)) % 4]
    cut["t_max"] = np.where(np.isfinite(t), t * factor, np.float32(1.0))
    t_cut, n_cut, _ = _check_once(name + " (segments)", intersect(cut), intersect_p(cut), pos, idx, cut, to_object, exact)
    assert 0 < n_cut < n_hit or n_hit == 0
    return n_hit, n_off


@pytest.mark.parametrize("split", [0, 1, 2, 3])
def test_oracle_aggregate_is_the_brute_force(split):
    """Every split method's tree (SAH, HLBVH, Middle, EqualCounts), leaves of 1 and 4 primitives."""
    for name, (pos, idx, rays) in _scenes().items():
        for max_prims in (1, 4):
            osc = oracle.OracleScene(_as_scene(pos, idx), max_prims_in_node=max_prims, split_method=split)
            n_hit, n_tied = _check(f"{name} split {split} max_prims {max_prims}", lambda r: osc.intersect(r)[0], lambda r: osc.intersect_p(r)[0],
                                   pos, idx, rays, exact=name == "cloud")
            assert n_hit > len(rays) // 10
            osc.close()


def test_oracle_instances_are_the_brute_force():
    """TransformedPrimitive::intersect (primitive.rs:136-159): every instance's triangles under its world-to-object matrix."""
    sc = scenes.instanced_scene(300, 12, extent=1.0, base_extent=0.4, tri_size=0.12)
    osc = oracle.OracleScene(sc)
    rays = scenes.random_rays(2000, 8, origin_extent=1.5)
    to_object = np.asarray(sc["instances"], dtype=np.float32)[:, 1].reshape(-1, 16)
    n_hit, _ = _check("instanced", lambda r: osc.intersect(r)[0], lambda r: osc.intersect_p(r)[0], sc["positions"], sc["indices"], rays, to_object)
    assert n_hit > 100
    osc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", [0, 1, 2])
def test_device_aggregate_is_the_brute_force(hip_ctx, traversal):
    """The library's builder, its 4-wide records (0), binary records with a stack (1) and stackless (2)."""
    hip_ctx.set_traversal(traversal)
    try:
        for name, (pos, idx, rays) in _scenes().items():
            for split in (0, 1):
                g = pbrt_hip.Scene(hip_ctx, _as_scene(pos, idx), split_method=split)
                _check(f"{name} traversal {traversal} split {split}", g.intersect, g.intersect_p, pos, idx, rays, exact=name == "cloud")
                g.close()
    finally:
        hip_ctx.set_traversal(0)


@pytest.mark.gpu
def test_device_instances_are_the_brute_force(hip_ctx):
    sc = scenes.instanced_scene(300, 12, extent=1.0, base_extent=0.4, tri_size=0.12)
    g = pbrt_hip.Scene(hip_ctx, sc)
    rays = scenes.random_rays(2000, 8, origin_extent=1.5)
    to_object = np.asarray(sc["instances"], dtype=np.float32)[:, 1].reshape(-1, 16)
    n_hit, _ = _check("instanced on the device", g.intersect, g.intersect_p, sc["positions"], sc["indices"], rays, to_object)
    assert n_hit > 100
    g.close()


def _brute_two_level(sc, rays):
    """The general top level (primitive.rs:105-159 beside :33-103) without any tree: every instance's object triangles under its
    world-to-object matrix, and the world-space triangles, all put to the triangle test; the smallest t, and who reaches it."""
    n = len(rays)
    best = np.full(n, np.inf, dtype=np.float32)
    prim, inst = np.full(n, -1, dtype=np.int32), np.full(n, -1, dtype=np.int32)
    to_object = np.asarray(sc["instances"], dtype=np.float32)[:, 1].reshape(-1, 16)
    io = np.asarray(sc["instance_object"])
    for k, obj in enumerate(sc["objects"]):
        sel = np.flatnonzero(io == k)
        if len(sel) == 0:
            continue
        t, p, i, _ = oracle.brute_force(obj["positions"], obj["indices"], rays, to_object[sel])
        closer = t < best
        best, prim, inst = np.where(closer, t, best), np.where(closer, p, prim), np.where(closer, sel[np.maximum(i, 0)], inst)
    w = sc["world"]
    if len(w["indices"]):
        t, p, _, _ = oracle.brute_force(w["positions"], w["indices"], rays)
        closer = t < best
        best, prim, inst = np.where(closer, t, best), np.where(closer, p, prim), np.where(closer, -1, inst)
    return best, prim, inst


def _check_two_level(name, hits, any_hit, sc, rays):
    t, prim, inst = _brute_two_level(sc, rays)
    found = hits["prim_id"] >= 0
    assert np.array_equal(found, np.isfinite(t)) and np.array_equal(any_hit.astype(bool), found), name
    tb, th = t[found].astype(np.float64), hits["t"][found].astype(np.float64)
    assert np.all(th >= tb) and np.all(th - tb <= 4 * 2.0 ** -23 * tb), (name, float(np.max((th - tb) / tb)))
    # clouds of separate triangles: no shared edges, the aggregate and the brute force name the same hit
    assert np.array_equal(hits["t"][found], t[found]) and np.array_equal(hits["prim_id"][found], prim[found]), name
    assert np.array_equal(hits["instance_id"][found], inst[found]), name
    assert (inst[found] >= 0).sum() > 100 and (inst[found] < 0).sum() > 100, name      # both kinds of primitive were hit
    return int(found.sum())


def test_oracle_general_two_level_scene_is_the_brute_force():
    sc = scenes.two_level_scene(30)
    osc = oracle.OracleScene(sc)
    rays = scenes.random_rays(6000, 9, origin_extent=3.0)
    _check_two_level("two-level", osc.intersect(rays)[0], osc.intersect_p(rays)[0], sc, rays)
    osc.close()


@pytest.mark.gpu
def test_device_general_two_level_scene_is_the_brute_force(hip_ctx):
    sc = scenes.two_level_scene(30)
    g = pbrt_hip.Scene(hip_ctx, sc)
    rays = scenes.random_rays(6000, 9, origin_extent=3.0)
    _check_two_level("two-level on the device", g.intersect(rays), g.intersect_p(rays), sc, rays)
    g.close()
