"""The 4-wide quantised records are the traversal that runs (not a silent use of the binary records), on every kind of
scene the library builds them for, and the three kernels agree: wide records (default), binary records (the
instrumented reference-count kernel, set_counting(1)) and the oracle's BVHAccel::intersect / intersect_p
(src/accelerators/bvh.rs:812-945) give the same hits bit for bit. Rays outside the range the filter's error bound covers
(wide_bvh.h: wide_ray_covered) are traced by the binary kernel in a follow-up launch, and are counted.
"""
import os

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


def _rays_into(lo, hi, n, seq):
    u = scenes.pcg32_float(seq, n * 6).reshape(n, 6)
    ext = (hi - lo).astype(np.float32)
    r = scenes.random_rays(n, seq + 1)
    r["o"] = (lo + (u[:, :3] * 1.6 - 0.3) * ext).astype(np.float32)
    aim = (lo + u[:, 3:6] * ext).astype(np.float32)
    d = aim - r["o"]
    d[np.all(d == 0, axis=1)] = (0.1, 0.2, 0.3)
    r["d"] = d.astype(np.float32)
    r["t_max"][::5] = np.float32(0.6)
    return r


def _three_way(ctx, gsc, osc, rays):
    """(wide hits, wide occlusion, wide counters) after asserting wide == binary == oracle."""
    ctx.set_counting(2)
    try:
        ctx.wide_counters(reset=True)
        hw, pw = gsc.intersect(rays), gsc.intersect_p(rays)
        wc = ctx.wide_counters(reset=True)
        ctx.set_counting(1)
        ctx.counters(reset=True)
        hb, pb = gsc.intersect(rays), gsc.intersect_p(rays)
    finally:
        ctx.set_counting(0)
    hd, pd = gsc.intersect(rays), gsc.intersect_p(rays)          # the uninstrumented wide kernels
    assert hw.tobytes() == hb.tobytes() == hd.tobytes()
    assert np.array_equal(pw, pb) and np.array_equal(pw, pd)
    cpu, _ = osc.intersect(rays)
    for f in ("prim_id", "t", "b0", "b1", "b2"):
        assert np.array_equal(hw[f], cpu[f]), f
    assert np.array_equal(pw, osc.intersect_p(rays)[0])
    return hw, pw, wc


@pytest.mark.parametrize("which", ["cloud", "cornell", "mixed", "instanced", "two_level", "device_built"])
def test_wide_records_are_what_runs(hip_ctx, which):
    kw = {}
    if which == "cloud":
        sc = scenes.random_triangles(60_000, seq=4, size=0.04)
    elif which == "cornell":
        sc = scenes.cornell_box()
    elif which == "mixed":
        sc = scenes.mixed_materials_scene()
    elif which == "instanced":
        sc = scenes.instanced_scene(n_base_tris=3000, n_instances=300)
    elif which == "two_level":
        sc = scenes.two_level_scene(n_instances=70)
    else:
        sc, kw = scenes.random_triangles(60_000, seq=4, size=0.04), dict(device_build=True)
    gsc = pbrt_hip.Scene(hip_ctx, sc, **kw)
    n_rec, why = gsc.wide_records()
    assert n_rec > 0, why
    if which == "device_built":
        osc = oracle.OracleScene(sc, split_method=pbrt_hip.SPLIT_HLBVH)
    else:
        osc = oracle.OracleScene(sc)
    root = osc.nodes()[0]
    rays = _rays_into(root["bmin"], root["bmax"], 80_000, 41)
    hits, occl, wc = _three_way(hip_ctx, gsc, osc, rays)
    assert (hits["prim_id"] >= 0).sum() > 2000 and occl.sum() > 2000
    assert wc["records"] > len(rays) and wc["triangles"] > 0        # the wide kernels did the work ...
    assert wc["special_rays"] < len(rays) // 50                     # ... for all but a few rays
    gsc.close()
    osc.close()


@pytest.mark.parametrize("device_build", [False, True])
def test_scene_far_from_the_origin_keeps_the_binary_records(hip_ctx, device_build):
    """at coordinates around 6e4 the byte-in-base displacement (256 ulp = a whole unit) would leave the 8-bit
    planes of small nodes filtering nothing; both builders (the device one over a host tree and over a device-built tree)
    count such records and decline the tree. The scene still traces exactly, over the binary records; moved by 60 only, it
    keeps its wide records and steps about as many of them per ray as the scene at the origin."""
    base = scenes.random_triangles(30_000, seq=8, size=0.03)
    steps = {}
    for off in (0.0, 60.0, 60000.0):
        sc = dict(base, positions=(base["positions"] + np.float32(off)).astype(np.float32))
        gsc = pbrt_hip.Scene(hip_ctx, sc, device_build=device_build)
        osc = oracle.OracleScene(sc, split_method=pbrt_hip.SPLIT_HLBVH) if device_build else oracle.OracleScene(sc)
        n_rec, why = gsc.wide_records()
        root = osc.nodes()[0]
        rays = _rays_into(root["bmin"], root["bmax"], 20_000, 5)
        hip_ctx.set_counting(2)
        hip_ctx.wide_counters(reset=True)
        got = gsc.intersect(rays)
        wc = hip_ctx.wide_counters(reset=True)
        hip_ctx.set_counting(0)
        cpu, _ = osc.intersect(rays)
        for f in ("prim_id", "t", "b0", "b1", "b2"):
            assert np.array_equal(got[f], cpu[f]), (off, f)
        if off > 1000.0 and n_rec < 0:
            assert "too far from the origin" in why, (n_rec, why)
            assert wc["records"] == 0
        elif off > 1000.0:
            # (the device-built HLBVH tree: leaves of four triangles, larger nodes, fewer than one coarse record in eight —
            # it keeps its records, and they must still filter)
            assert device_build and wc["records"] / len(rays) <= 3.0 * steps[0.0], (wc, steps)
        else:
            assert n_rec > 0, why
            steps[off] = wc["records"] / len(rays)
        gsc.close()
        osc.close()
    assert steps[60.0] <= 1.15 * steps[0.0], steps


def test_rays_outside_the_filters_range_take_the_binary_records(hip_ctx):
    """Axis-parallel directions (infinite reciprocals), denormal-small components, far origins: the wide kernel lists them,
    a follow-up launch of the binary kernel traces them; the count is exactly the number wide_ray_covered rejects."""
    sc = scenes.random_triangles(20_000, seq=3, size=0.05)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    osc = oracle.OracleScene(sc)
    n = 40_000
    rays = _rays_into(np.float32([-1, -1, -1]), np.float32([1, 1, 1]), n, 57)
    k = np.arange(n)
    a = k % 3
    par = k % 4 == 0
    rays["d"][par, a[par]] = 0.0                                   # 1/0 = inf
    tiny = k % 4 == 1
    rays["d"][tiny, a[tiny]] = np.float32(1e-30)                   # |1/d| > 2^40
    far = k % 16 == 2
    rays["o"][far, a[far]] = np.float32(-3e7)                      # |o| > 2^24, aimed back at the scene
    rays["d"][far] = -rays["o"][far] + rays["d"][far] * np.float32(0.1)
    huge = k % 16 == 3
    rays["d"][huge] *= np.float32(1e15)                            # |1/d| < 2^-40 on every axis
    with np.errstate(divide="ignore"):
        inv = np.abs(np.float32(1) / rays["d"])
    covered = (np.all((inv >= np.float32(2.0 ** -40)) & (inv <= np.float32(2.0 ** 40)), axis=1)
               & np.all(np.abs(rays["o"]) <= np.float32(2.0 ** 24), axis=1))
    assert 0.3 * n < (~covered).sum() < 0.7 * n
    hits, occl, wc = _three_way(hip_ctx, gsc, osc, rays)
    assert wc["special_rays"] == 2 * int((~covered).sum())         # once per launch: intersect and intersect_p
    assert (hits["prim_id"][~covered] >= 0).sum() > 500 and (hits["prim_id"][covered] >= 0).sum() > 500
    gsc.close()
    osc.close()


def test_wide_traversal_fetches_fewer_records_than_the_reference_loop_tests_boxes(hip_ctx):
    """What the wide records are for: per ray, records fetched (48 B each) against the reference loop's box tests (32 B
    nodes). The ratio on the bench's kind of scene is what DESIGN.md's request count rests on."""
    sc = scenes.random_triangles(200_000, seq=1, size=0.02)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    rays = scenes.random_rays(200_000, 5, origin_extent=1.2)
    hip_ctx.set_counting(2)
    try:
        hip_ctx.wide_counters(reset=True)
        gsc.intersect(rays)
        wc = hip_ctx.wide_counters(reset=True)
        hip_ctx.set_counting(1)
        hip_ctx.counters(reset=True)
        gsc.intersect(rays)
        bc = hip_ctx.counters(reset=True)
    finally:
        hip_ctx.set_counting(0)
    assert wc["records"] * 2.2 < bc["node_tests"]
    assert wc["triangles"] <= bc["prim_tests"] * 1.1 + 100     # leaves pass the exact box first, so few extra triangles
    gsc.close()


@pytest.mark.parametrize("split,max_prims", [(pbrt_hip.SPLIT_SAH, 4), (pbrt_hip.SPLIT_MIDDLE, 1), (pbrt_hip.SPLIT_EQUAL_COUNTS, 2), (pbrt_hip.SPLIT_HLBVH, 4)])
def test_visiting_order_decides_ties(hip_ctx, split, max_prims):
    """Every triangle three times, under different primitive numbers: all copies give the same t, and the reference keeps
    the one its traversal reaches first (Triangle::intersect_test needs t < t_max, triangle.rs:127-131). The primitive
    number returned is therefore a direct read-out of the visiting order of leaves and of triangles inside them."""
    base = scenes.random_triangles(4000, seq=12, size=0.15)
    n = len(base["indices"])
    perm = np.argsort(scenes.pcg32_float(3, 3 * n), kind="stable").astype(np.int64)   # the copies are not neighbours in the input
    sc = dict(base, indices=np.ascontiguousarray(np.tile(base["indices"], (3, 1))[perm]),
              tri_material=np.ascontiguousarray(np.tile(base["tri_material"], 3)[perm]),
              tri_light=np.ascontiguousarray(np.tile(base["tri_light"], 3)[perm]))
    gsc = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=max_prims, split_method=split)
    assert gsc.wide_records()[0] > 0
    osc = oracle.OracleScene(sc, max_prims_in_node=max_prims, split_method=split)
    rays = scenes.random_rays(60_000, 8, origin_extent=1.3)
    hits, occl, wc = _three_way(hip_ctx, gsc, osc, rays)
    hit = hits["prim_id"] >= 0
    assert hit.sum() > 10_000
    copies = perm[hits["prim_id"][hit]] // n        # which of the three copies won: all three occur
    assert len(np.unique(copies)) == 3
    gsc.close()
    osc.close()


@pytest.mark.parametrize("tree", ["built_on_device", "from_host"])
@pytest.mark.parametrize("which", ["cloud", "cornell", "mixed", "one_leaf"])
def test_device_builder_of_the_wide_records_equals_the_host_builder(hip_ctx, which, tree):
    """Single-level scenes get their wide records on the device (csrc/wide_gpu.hip, level by level): over a tree built there
    (pbrt_hip_scene_create_hlbvh) and over a tree the caller built (its flat nodes go up once). PBRT_WIDE_BUILD_HOST (pbrt_hip_context_set_wide_build)
    runs the host builder (csrc/host_wide.cpp) instead. Same source for the arithmetic (csrc/wide_build.h), same order:
    the three arrays are equal byte for byte, and so is everything traced."""
    kw = dict(device_build=True) if tree == "built_on_device" else dict(split_method=pbrt_hip.SPLIT_SAH)
    if which == "cloud":
        sc = scenes.random_triangles(150_000, seq=21, size=0.03)
    elif which == "cornell":
        sc = scenes.cornell_box()
    elif which == "mixed":
        sc = scenes.mixed_materials_scene()
    else:
        base = scenes.cornell_box()
        sc = dict(base, indices=base["indices"][:1].copy(), tri_material=base["tri_material"][:1].copy(), tri_light=base["tri_light"][:1].copy())
        sc = scenes.with_lights(sc, [scenes.point_light((278.0, 400.0, 278.0), (1e5, 1e5, 1e5))], keep_existing=False)
    n = len(sc["indices"])
    g_dev = pbrt_hip.Scene(hip_ctx, sc, **kw)
    hip_ctx.set_wide_build(pbrt_hip.WIDE_BUILD_HOST)
    try:
        g_host = pbrt_hip.Scene(hip_ctx, sc, **kw)
        hip_ctx.set_wide_build(pbrt_hip.WIDE_BUILD_NONE)
        g_none = pbrt_hip.Scene(hip_ctx, sc, **kw)
    finally:
        hip_ctx.set_wide_build(pbrt_hip.WIDE_BUILD_DEVICE)      # the context is shared by the session
    assert g_none.wide_records() == (-1, "disabled by PBRT_WIDE_BUILD_NONE")
    assert g_dev.wide_records() == g_host.wide_records() and g_dev.wide_records()[0] >= 0, (g_dev.wide_records(), g_host.wide_records())
    for a, b, what in zip(g_dev.debug_wide_export(n), g_host.debug_wide_export(n), ("records", "triangles", "leaf boxes")):
        assert a.tobytes() == b.tobytes(), what
    if which == "one_leaf":
        assert g_dev.wide_records()[0] == 0
    rays = scenes.random_rays(50_000, 77, origin_extent=600.0 if which in ("cornell", "one_leaf") else 1.3)
    assert g_dev.intersect(rays).tobytes() == g_host.intersect(rays).tobytes() == g_none.intersect(rays).tobytes()
    g_none.close()
    g_dev.close()
    g_host.close()


@pytest.mark.parametrize("instanced", [False, True])
def test_stacks_deeper_than_the_lds_part(hip_ctx, instanced):
    """Large overlapping triangles, one per leaf: a ray through the middle passes nearly every box, up to three children
    are pushed per record level, and the per-lane stack outgrows its 12 LDS entries (the spill slab in global memory and the
    predicated push path take over from the unconditional LDS stores). Same hits as the binary kernel and the oracle."""
    if instanced:
        sc = scenes.instanced_scene(n_base_tris=2000, n_instances=40, tri_size=0.25, base_extent=0.3, extent=0.6)
        kw = {}
    else:
        sc = scenes.random_triangles(3000, seq=5, size=0.4)
        kw = dict(max_prims_in_node=1, split_method=pbrt_hip.SPLIT_MIDDLE)
    gsc = pbrt_hip.Scene(hip_ctx, sc, **kw)
    assert gsc.wide_records()[0] > 0
    if not instanced:
        # the builder's bound on the stack depth, recomputed from the records: k children leave k - 1 on the stack
        rec = gsc.debug_wide_export(len(sc["indices"]))[0]
        m = np.stack([rec[:, 0] & 0xff, rec[:, 1] & 0xff, rec[:, 2] & 0xff, rec[:, 3] >> 24], axis=1).astype(np.int64)
        need = np.zeros(len(rec), dtype=np.int64)
        for w in range(len(rec) - 1, -1, -1):          # children have larger indices (breadth-first layout)
            live = m[w] != 0xff
            kids = [int(rec[w, 10]) + int(v & 3) for v in m[w][live & ((m[w] & 0x80) != 0)]]
            need[w] = int(live.sum()) - 1 + (max(need[k] for k in kids) if kids else 0)
        assert need[0] > 14, need[0]
    osc = oracle.OracleScene(sc, **kw)
    root = osc.nodes()[0]
    rays = _rays_into(root["bmin"], root["bmax"], 30_000, 91)
    rays["t_max"][:] = np.inf
    hits, occl, wc = _three_way(hip_ctx, gsc, osc, rays)
    assert wc["records"] > 40 * len(rays)          # long walks: tens of records per ray and launch
    gsc.close()
    osc.close()


def test_adversarial_meshes_and_rays_through_vertices_and_edges(hip_ctx):
    """Hypothesis-generated small meshes (coincident vertices, zero-area triangles, repeated triangles, large and tiny
    coordinates) under every host split method, with rays aimed exactly at vertices and edge midpoints (the cases where
    the watertight test's ties and the box planes of single-triangle leaves meet), axis-parallel rays among them (the
    hand-over to the binary kernel). Whatever the builder decides (records, or a refusal with a reason), the answers are
    the oracle's."""
    from hypothesis import given, settings, strategies as st
    from test_property_host import meshes

    @settings(max_examples=int(os.environ.get("PB_HYP_EXAMPLES", "400")), deadline=None)
    @given(mesh=meshes(), max_prims=st.sampled_from([1, 2, 4]), split=st.sampled_from([0, 1, 2, 3]))
    def check(mesh, max_prims, split):
        verts, idx = mesh
        sc = dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
                  materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                  tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))
        tri = verts[idx]                                                    # (n, 3, 3)
        targets = np.concatenate([tri.reshape(-1, 3), (tri[:, 0] + tri[:, 1]) * np.float32(0.5), tri.mean(axis=1)])
        n = 3 * len(targets)
        rays = scenes.random_rays(n, 7, origin_extent=2.0)
        tgt = targets[np.arange(n) % len(targets)]
        scale = np.maximum(np.abs(tgt).max(axis=1, keepdims=True), 1.0).astype(np.float32)
        rays["o"] = (tgt + rays["o"] * scale).astype(np.float32)           # origins around the target, at its own scale
        rays["d"] = (tgt - rays["o"]).astype(np.float32)
        k = np.arange(n)
        par = k % 5 == 0
        rays["d"][par, k[par] % 3] = 0.0                                    # axis-parallel: 1 / 0 in the slab test
        rays["d"][np.all(rays["d"] == 0, axis=1)] = (0.0, 0.0, 1.0)
        finite = np.isfinite(rays["o"]).all(axis=1) & np.isfinite(rays["d"]).all(axis=1)
        rays = np.ascontiguousarray(rays[finite])
        if len(rays) == 0:
            return
        osc = oracle.OracleScene(sc, max_prims, split)
        gsc = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=max_prims, split_method=split)
        try:
            n_rec, why = gsc.wide_records()
            assert n_rec >= 0 or why                                         # records, or a stated reason
            cpu, _ = osc.intersect(rays)
            gpu = gsc.intersect(rays)
            for f in ("prim_id", "t", "b0", "b1", "b2"):
                assert np.array_equal(gpu[f], cpu[f]), (f, n_rec, why)
            assert np.array_equal(gsc.intersect_p(rays), osc.intersect_p(rays)[0])
        finally:
            gsc.close()
            osc.close()

    check()


def test_state_stream_probe_reports_what_its_lanes_ask_for(hip_ctx):
    """pbrt_hip_probe_state_stream (the calibration probe behind bench.py's roofline.shade, profiles/r04_fetch_size_calibration_shade.txt):
    the byte counts it reports are the per-path widths of include/pbrt_hip.h times the queue length, for every part mask."""
    n, density = 1 << 20, 0.7
    keep = -(-1024 * 700 // 1000)
    q = n // 1024 * keep
    widths = {1: 9 * 16, 2: 2 * 32, 4: 16, 8: 48}
    for parts in (1, 2, 4, 8, 15, 16, 31):
        rd, wr, ms = hip_ctx.probe_state_stream(n, density, 1 << 20, parts)
        assert rd == q * (4 + sum(w for bit, w in widths.items() if parts & bit)), parts
        assert wr == (q * (5 * 16 + 3 * 32 + 16 + 7 * 4) if parts & 16 else 0), parts
        assert ms > 0
    with pytest.raises(pbrt_hip.PbrtHipError):
        hip_ctx.probe_state_stream(100, 0.5, 1 << 20, 1)


@pytest.mark.parametrize("which", ["cloud", "cornell", "instanced", "two_level", "device_built", "vertex_data"])
def test_packed_and_line_aligned_layouts_are_the_same_scene(hip_ctx, which):
    """pbrt_hip_context_set_wide_layout: the 4-wide records and their triangles 48 bytes apart (packed) or one 64-byte line each
    (WideTrees::vec_stride 3 / 4; the builders always produce the packed form, the scene spreads it). Every kind of scene both
    ways: the strides are what was asked for, the exported bytes are the same, hits / occlusion / counters are the same bit for
    bit and equal the oracle's, a render is the same film — and the flags the shading data refreshes in the wide-order triangles
    (pbrt_hip_scene_set_shading_data -> k_wide_refresh_flags) land in the right place at either stride."""
    kw = {}
    if which == "cloud":
        sc = scenes.random_triangles(60_000, seq=4, size=0.04)
    elif which == "cornell":
        sc = scenes.cornell_box()
    elif which == "instanced":
        sc = scenes.instanced_scene(n_base_tris=3000, n_instances=300)
    elif which == "two_level":
        sc = scenes.two_level_scene(n_instances=70)
    elif which == "vertex_data":
        sc = scenes.with_vertex_shading(scenes.random_triangles(20_000, seq=6, size=0.05), seq=9)
    else:
        sc, kw = scenes.random_triangles(60_000, seq=4, size=0.04), dict(device_build=True)
    got = {}
    try:
        for layout, stride in ((pbrt_hip.WIDE_LAYOUT_PACKED, 48), (pbrt_hip.WIDE_LAYOUT_LINES, 64)):
            hip_ctx.set_wide_layout(layout)
            g = pbrt_hip.Scene(hip_ctx, sc, **kw)
            # (two-level scenes are always packed: their kernels keep a compile-time stride)
            assert g.wide_records()[0] > 0 and g.wide_stride() == (48 if which in ("instanced", "two_level") else stride)
            got[stride] = g
    finally:
        hip_ctx.set_wide_layout(pbrt_hip.WIDE_LAYOUT_AUTO)        # the context is shared by the session
    a, b = got[48], got[64]
    assert a.wide_records() == b.wide_records()
    if which not in ("instanced", "two_level"):
        n = len(sc["indices"])
        for x, y, what in zip(a.debug_wide_export(n), b.debug_wide_export(n), ("records", "triangles", "leaf boxes")):
            assert x.tobytes() == y.tobytes(), what
    if which == "device_built":
        osc = oracle.OracleScene(sc, split_method=pbrt_hip.SPLIT_HLBVH)
    else:
        osc = oracle.OracleScene(sc, normals=sc.get("normals"), uvs=sc.get("uvs"), tangents=sc.get("tangents"))
    root = osc.nodes()[0]
    rays = _rays_into(root["bmin"], root["bmax"], 60_000, 43)
    ha, pa, wca = _three_way(hip_ctx, a, osc, rays)
    hb, pb, wcb = _three_way(hip_ctx, b, osc, rays)
    assert ha.tobytes() == hb.tobytes() and np.array_equal(pa, pb) and wca == wcb and (ha["prim_id"] >= 0).sum() > 1000
    if which in ("cornell", "vertex_data", "instanced"):
        cam = {"cornell": scenes.cornell_camera, "instanced": scenes.instanced_camera}.get(which, scenes.random_triangles_camera)(64, 48)
        fa, sta = a.render(cam, 64, 48, 4, max_depth=6, seed=5)
        fb, stb = b.render(cam, 64, 48, 4, max_depth=6, seed=5)
        assert fa.tobytes() == fb.tobytes() and sta["rays_closest"] == stb["rays_closest"] and sta["rays_shadow"] == stb["rays_shadow"]
    osc.close()
    a.close()
    b.close()


def test_wide_layout_follows_the_size_of_the_tree(hip_ctx):
    """PBRT_WIDE_LAYOUT_AUTO: records + triangles that fit 8 MiB (a tree L2 can hold) stay packed — padding would only cost them
    cache; larger ones get one line per record (no record straddles two 64-byte lines: profiles/r05_line_aligned.txt)."""
    small = pbrt_hip.Scene(hip_ctx, scenes.random_triangles(50_000, seq=4, size=0.04))     # ~24 k records + 50 k triangles: 3.6 MB
    large = pbrt_hip.Scene(hip_ctx, scenes.random_triangles(200_000, seq=4, size=0.03))    # ~98 k records + 200 k triangles: 14 MB
    inst = pbrt_hip.Scene(hip_ctx, scenes.instanced_scene(n_base_tris=3000, n_instances=300))
    assert (small.wide_stride(), large.wide_stride(), inst.wide_stride()) == (48, 64, 48)
    with pytest.raises(pbrt_hip.PbrtHipError, match="PBRT_WIDE_LAYOUT"):
        hip_ctx.set_wide_layout(3)
    hip_ctx.set_wide_build(pbrt_hip.WIDE_BUILD_NONE)
    try:
        none = pbrt_hip.Scene(hip_ctx, scenes.cornell_box())
    finally:
        hip_ctx.set_wide_build(pbrt_hip.WIDE_BUILD_DEVICE)
    assert none.wide_stride() == 0
    for g in (small, large, inst, none):
        g.close()
