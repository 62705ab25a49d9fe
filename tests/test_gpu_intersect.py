"""GPU parity: batch Scene::intersect / intersect_p through the C ABI vs the CPU oracle.

Bar: bit-exact (hit, prim_id, t, b0, b1, b2) — the kernels evaluate the same IEEE operations in
the same order as the oracle (src/accelerators/bvh.rs:828-932, src/shapes/triangle.rs:74-158).
"""
import os

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


def _assert_hits_equal(gpu, cpu):
    assert np.array_equal(gpu["prim_id"], cpu["prim_id"])
    hit = cpu["prim_id"] >= 0
    for f in ("t", "b0", "b1", "b2"):
        assert np.array_equal(gpu[f][hit].view(np.uint32), cpu[f][hit].view(np.uint32)), f
    assert np.all(np.isinf(gpu["t"][~hit]))


def _scene_rays(sc, n, seq, extent):
    rays = scenes.random_rays(n, seq, origin_extent=extent)
    # a quarter of the rays get a finite t_max (shadow-ray style upper bounds)
    rays["t_max"][::4] = np.float32(0.75)
    return rays


@pytest.mark.parametrize("name,n_rays", [("cornell", 50_000), ("rand20k", 100_000), ("mixed", 100_000)])
def test_intersect_parity(hip_ctx, name, n_rays):
    sc = {"cornell": scenes.cornell_box, "rand20k": lambda: scenes.random_triangles(20_000, seq=3, size=0.05),
          "mixed": scenes.mixed_materials_scene}[name]()
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    rays = _scene_rays(sc, n_rays, 11, 600.0 if name == "cornell" else 1.5)
    if name == "cornell":
        rays["o"] = np.abs(rays["o"]) * np.float32(0.9)  # inside the box
        rays["t_max"][::4] = np.float32(300.0)
    cpu, ctr = osc.intersect(rays)
    gpu = gsc.intersect(rays)
    _assert_hits_equal(gpu, cpu)
    assert (cpu["prim_id"] >= 0).sum() > n_rays // 20
    cpu_p, _ = osc.intersect_p(rays)
    gpu_p = gsc.intersect_p(rays)
    assert np.array_equal(gpu_p, cpu_p)
    gsc.close()
    osc.close()


def test_intersect_edge_cases(hip_ctx):
    sc = scenes.cornell_box()
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    # empty batch
    assert len(gsc.intersect(np.zeros(0, dtype=scenes.RAY_DTYPE))) == 0
    assert len(gsc.intersect_p(np.zeros(0, dtype=scenes.RAY_DTYPE))) == 0
    # axis-parallel rays (zero direction components -> infinite inv_dir), rays on shared edges and
    # vertices of the floor quad, rays starting on a surface, zero-length t_max, a ragged count
    rays = np.zeros(777, dtype=scenes.RAY_DTYPE)
    rays["t_max"] = np.inf
    g = scenes.pcg32_float(5, 777 * 3).reshape(777, 3)
    rays["o"] = g * np.float32(555.0)
    dirs = np.array([[0, -1, 0], [0, 1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [1, -1, 0], [0, -1, 1]],
                    dtype=np.float32)
    rays["d"] = dirs[np.arange(777) % 8]
    rays["o"][:40, 0] = rays["o"][:40, 2]          # above the floor diagonal (shared edge of the two triangles)
    rays["d"][:40] = (0, -1, 0)
    rays["o"][40:44] = [(0, 300, 0), (555, 300, 555), (0, 300, 555), (555, 300, 0)]  # above the corners
    rays["d"][40:44] = (0, -1, 0)
    rays["o"][44:60, 1] = 0.0                        # origins exactly on the floor
    rays["t_max"][60:70] = 0.0
    cpu, _ = osc.intersect(rays)
    gpu = gsc.intersect(rays)
    _assert_hits_equal(gpu, cpu)
    assert np.array_equal(gsc.intersect_p(rays), osc.intersect_p(rays)[0])
    gsc.close()
    osc.close()


def test_single_triangle_scene(hip_ctx):
    """A tree that is a single leaf (root reference is a leaf)."""
    sc = scenes.furnace_scene()
    sc = dict(sc, positions=sc["positions"][:3].copy(), indices=sc["indices"][:1].copy(),
              tri_material=sc["tri_material"][:1].copy(), tri_light=sc["tri_light"][:1].copy())
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    rays = scenes.random_rays(5000, 9, origin_extent=50.0)
    rays["o"][:, 1] = np.abs(rays["o"][:, 1]) + 1
    rays["d"][:, 1] = -np.abs(rays["d"][:, 1]) - np.float32(0.1)
    cpu, _ = osc.intersect(rays)
    _assert_hits_equal(gsc.intersect(rays), cpu)
    assert (cpu["prim_id"] >= 0).any()
    gsc.close()
    osc.close()


def test_instrumented_counts_match_reference_loops(hip_ctx):
    """The instrumented kernel reports exactly the box / triangle test counts of the reference's
    traversal loops (as counted by the oracle) — the inputs of the roofline's algorithmic bytes."""
    sc = scenes.random_triangles(20_000, seq=3, size=0.05)
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    rays = _scene_rays(sc, 60_000, 21, 1.5)
    hip_ctx.set_counting(True)
    try:
        hip_ctx.counters(reset=True)
        gpu = gsc.intersect(rays)
        c_closest = hip_ctx.counters(reset=True)
        gpu_p = gsc.intersect_p(rays)
        c_any = hip_ctx.counters(reset=True)
    finally:
        hip_ctx.set_counting(False)
    cpu, ctr = osc.intersect(rays)
    cpu_p, ctr_p = osc.intersect_p(rays)
    _assert_hits_equal(gpu, cpu)
    assert np.array_equal(gpu_p, cpu_p)
    assert c_closest == ctr
    assert c_any == ctr_p
    gsc.close()
    osc.close()


def test_general_two_level_scene_parity(hip_ctx):
    """The general TransformedPrimitive scene (src/core/primitive.rs:105-159): three different object aggregates,
    instances cycling through them, and world-space GeometricPrimitives (a floor and an emitting quad) beside the instances
    in the same top-level BVHAccel. Bit-exact hits (triangle index inside its object, instance id, -1 for the world
    triangles), any-hit flags and the reference-loop counters, top-level triangle tests included."""
    sc = scenes.two_level_scene()
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    assert gsc.tlas_nodes.tobytes() == osc.nodes().tobytes() and np.array_equal(gsc.tlas_order, osc.prim_order())
    for k, (nodes, order) in enumerate(gsc.object_trees):
        on, oo = osc.object_tree(k)
        assert nodes.tobytes() == on.tobytes() and np.array_equal(order, oo)
    rays = _scene_rays(sc, 120_000, 41, 3.0)
    rays["t_max"][::4] = np.float32(2.5)
    cpu, ctr = osc.intersect(rays)
    cpu_p, ctr_p = osc.intersect_p(rays)
    gpu = gsc.intersect(rays)
    _assert_hits_equal(gpu, cpu)
    assert np.array_equal(gpu["instance_id"], cpu["instance_id"])
    hit = cpu["prim_id"] >= 0
    inst = sc["instance_object"][np.maximum(cpu["instance_id"], 0)]
    for k in range(3):          # every object aggregate is hit, and so are the world-space triangles
        assert (hit & (cpu["instance_id"] >= 0) & (inst == k)).sum() > 2000
    world = hit & (cpu["instance_id"] < 0)
    assert world.sum() > 5000 and set(np.unique(cpu["prim_id"][world])) == {0, 1, 2, 3}
    assert np.array_equal(gsc.intersect_p(rays), cpu_p)
    hip_ctx.set_counting(True)
    try:
        hip_ctx.counters(reset=True)
        gsc.intersect(rays)
        c = hip_ctx.counters(reset=True)
        gsc.intersect_p(rays)
        c_p = hip_ctx.counters(reset=True)
    finally:
        hip_ctx.set_counting(False)
    assert c == ctr and c_p == ctr_p
    gsc.close()
    osc.close()


def test_instanced_scene_parity(hip_ctx):
    """Two-level traversal (TransformedPrimitive instances, src/core/primitive.rs:136-159) vs the oracle:
    bit-exact hits incl. the instance id, any-hit flags, and the reference-loop counters."""
    sc = scenes.instanced_scene(3000, 60, extent=1.5)
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    assert gsc.tlas_nodes.tobytes() == osc.nodes().tobytes() and np.array_equal(gsc.tlas_order, osc.prim_order())
    bn, bo = osc.blas()
    assert gsc.nodes.tobytes() == bn.tobytes() and np.array_equal(gsc.prim_order, bo)
    rays = _scene_rays(sc, 80_000, 31, 2.5)
    cpu, ctr = osc.intersect(rays)
    cpu_p, ctr_p = osc.intersect_p(rays)
    gpu = gsc.intersect(rays)
    _assert_hits_equal(gpu, cpu)
    assert np.array_equal(gpu["instance_id"], cpu["instance_id"])
    assert (cpu["instance_id"] >= 0).sum() > 5000
    assert np.array_equal(gsc.intersect_p(rays), cpu_p)
    hip_ctx.set_counting(True)
    try:
        hip_ctx.counters(reset=True)
        g2 = gsc.intersect(rays)
        c0 = hip_ctx.counters(reset=True)
        gsc.intersect_p(rays)
        c1 = hip_ctx.counters(reset=True)
    finally:
        hip_ctx.set_counting(False)
    _assert_hits_equal(g2, cpu)
    assert c0 == ctr and c1 == ctr_p
    gsc.close()
    osc.close()


@pytest.mark.parametrize("split", [pbrt_hip.SPLIT_HLBVH, pbrt_hip.SPLIT_MIDDLE, pbrt_hip.SPLIT_EQUAL_COUNTS])
def test_other_split_methods(hip_ctx, split):
    """Trees from the other BVHAccel split methods (bvh.rs:200-205), max 16 primitives per leaf."""
    sc = scenes.random_triangles(20_000, seq=8, size=0.05)
    osc = oracle.OracleScene(sc, 16, split)
    gsc = pbrt_hip.Scene(hip_ctx, sc, 16, split)
    assert gsc.nodes.tobytes() == osc.nodes().tobytes()
    rays = _scene_rays(sc, 40_000, 12, 1.5)
    cpu, _ = osc.intersect(rays)
    _assert_hits_equal(gsc.intersect(rays), cpu)
    assert np.array_equal(gsc.intersect_p(rays), osc.intersect_p(rays)[0])
    gsc.close()
    osc.close()


@pytest.mark.parametrize("n,max_prims,seq", [(1, 4, 1), (3, 4, 2), (1000, 4, 3), (50_000, 4, 4), (50_000, 1, 5),
                                             (200_000, 8, 6), (20_000, 255, 7)])
def test_gpu_hlbvh_build_equals_host_and_oracle(hip_ctx, n, max_prims, seq):
    """GPU HLBVH (bvh.rs:475-568 + emit_lbvh / build_upper_sah / flatten): byte-identical to the host builder
    and to the oracle's BVHAccel::new(HLBVH)."""
    sc = scenes.random_triangles(n, seq=seq, size=0.05)
    nodes_h, order_h = pbrt_hip.bvh_build(sc["positions"], sc["indices"], max_prims, pbrt_hip.SPLIT_HLBVH)
    nodes_g, order_g, ms = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, sc["positions"], sc["indices"], max_prims)
    assert np.array_equal(order_g, order_h)
    assert nodes_g.tobytes() == nodes_h.tobytes()
    osc = oracle.OracleScene(sc, max_prims_in_node=max_prims, split_method=1)
    assert osc.nodes().tobytes() == nodes_g.tobytes()
    osc.close()


def test_gpu_hlbvh_clustered_and_duplicate_centroids(hip_ctx):
    """Few treelets (geometry in one corner), many identical Morton codes (bit == -1 leaves), Cornell box."""
    sc = scenes.random_triangles(30_000, seq=9, size=0.02)
    pos = sc["positions"].copy()
    pos[: pos.shape[0] // 2] *= np.float32(1e-3)             # half of the mesh collapses into one Morton cell
    nodes_h, order_h = pbrt_hip.bvh_build(pos, sc["indices"], 4, pbrt_hip.SPLIT_HLBVH)
    nodes_g, order_g, _ = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, pos, sc["indices"], 4)
    assert np.array_equal(order_g, order_h) and nodes_g.tobytes() == nodes_h.tobytes()
    dup = np.tile(sc["indices"][:7], (40, 1))                # 40 copies of 7 triangles: equal centroids
    nodes_h, order_h = pbrt_hip.bvh_build(sc["positions"], dup, 4, pbrt_hip.SPLIT_HLBVH)
    nodes_g, order_g, _ = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, sc["positions"], dup, 4)
    assert np.array_equal(order_g, order_h) and nodes_g.tobytes() == nodes_h.tobytes()
    cb = scenes.cornell_box()
    nodes_h, order_h = pbrt_hip.bvh_build(cb["positions"], cb["indices"], 4, pbrt_hip.SPLIT_HLBVH)
    nodes_g, order_g, _ = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, cb["positions"], cb["indices"], 4)
    assert np.array_equal(order_g, order_h) and nodes_g.tobytes() == nodes_h.tobytes()


def test_gpu_hlbvh_1m_triangles_and_trace(hip_ctx):
    """Config-3 mesh: GPU-built tree == host tree, and a scene over it answers ray queries like the SAH scene."""
    sc = scenes.random_triangles(1_000_000, seq=1)
    nodes_h, order_h = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_HLBVH)
    nodes_g, order_g, ms = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, sc["positions"], sc["indices"], 4)
    assert np.array_equal(order_g, order_h) and nodes_g.tobytes() == nodes_h.tobytes()
    assert 0.0 < ms < 200.0
    rays = scenes.random_rays(200_000, 5, origin_extent=1.2)
    g1 = pbrt_hip.Scene(hip_ctx, sc, bvh=(nodes_g, order_g))
    g2 = pbrt_hip.Scene(hip_ctx, sc)
    h1, h2 = g1.intersect(rays), g2.intersect(rays)
    assert np.array_equal(h1["prim_id"], h2["prim_id"]) and np.array_equal(h1["t"], h2["t"])
    g1.close()
    g2.close()


@pytest.mark.parametrize("which", ["cloud", "cornell", "mixed"])
def test_scene_create_hlbvh_on_device(hip_ctx, which):
    """pbrt_hip_scene_create_hlbvh: tree built and re-laid out on the device == the scene over the host HLBVH
    tree: same hits bit for bit, same reference-loop test counts, same rendered film (area-light slots included)."""
    if which == "cloud":
        sc, cam, w, h = scenes.random_triangles(120_000, seq=11, size=0.03), scenes.random_triangles_camera(64, 48), 64, 48
    elif which == "cornell":
        sc, cam, w, h = scenes.cornell_box(), scenes.cornell_camera(48, 48), 48, 48
    else:
        sc, cam, w, h = scenes.mixed_materials_scene(), scenes.random_triangles_camera(64, 48), 64, 48
    g_dev = pbrt_hip.Scene(hip_ctx, sc, device_build=True)
    g_host = pbrt_hip.Scene(hip_ctx, sc, split_method=pbrt_hip.SPLIT_HLBVH)
    assert g_dev.build_ms > 0.0 and g_dev.layout_ms > 0.0
    rays = scenes.random_rays(100_000, 21, origin_extent=1.2)
    hip_ctx.set_counting(True)
    try:
        res = []
        for g in (g_dev, g_host):
            hip_ctx.counters(reset=True)
            hits = g.intersect(rays)
            occl = g.intersect_p(rays)
            res.append((hits, occl, hip_ctx.counters(reset=True)))
    finally:
        hip_ctx.set_counting(False)
    assert res[0][0].tobytes() == res[1][0].tobytes() and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]
    f_dev, st_dev = g_dev.render(cam, w, h, 4, max_depth=5, seed=3)
    f_host, st_host = g_host.render(cam, w, h, 4, max_depth=5, seed=3)
    assert f_dev.tobytes() == f_host.tobytes() and st_dev["rays_shadow"] == st_host["rays_shadow"]
    g_dev.close()
    g_host.close()


def test_sphere_intersection_parity(hip_ctx):
    """Ray batches against spheres + triangles: same hit primitive and bit-identical t as the oracle's
    Sphere::intersect_test (EFloat quadratic), same any-hit answers, same reference-loop test counts, same tree."""
    sc = scenes.mixed_materials_scene()
    u = scenes.pcg32_float(13, 40 * 4).reshape(40, 4)
    sph = np.zeros((40, 8), dtype=np.float32)
    sph[:, :3], sph[:, 3], sph[:, 5] = u[:, :3] * 1.6 - 0.8, 0.03 + 0.25 * u[:, 3], -1
    sc["spheres"] = sph
    osc = oracle.OracleScene(sc)
    gsc = pbrt_hip.Scene(hip_ctx, sc)
    assert osc.nodes().tobytes() == gsc.nodes.tobytes() and np.array_equal(osc.prim_order(), gsc.prim_order)
    rays = scenes.random_rays(300_000, 31, origin_extent=1.5)
    rays["t_max"][:1000] = 0.5                                   # some short rays
    inside = rays[:2000].copy()
    inside["o"] = sph[np.arange(2000) % 40, :3]                  # rays starting at sphere centres (second root is the hit)
    rays = np.concatenate([rays, inside])
    cpu, st_c = osc.intersect(rays)
    hip_ctx.set_counting(True)
    hip_ctx.counters(reset=True)
    gpu = gsc.intersect(rays)
    c_g = hip_ctx.counters(reset=True)
    hip_ctx.set_counting(False)
    assert np.array_equal(gpu["prim_id"], cpu["prim_id"]) and np.array_equal(gpu["t"], cpu["t"])
    n_tris = sc["indices"].shape[0]
    tri_hit = (cpu["prim_id"] >= 0) & (cpu["prim_id"] < n_tris)
    for f in ("b0", "b1", "b2"):
        assert np.array_equal(gpu[f][tri_hit], cpu[f][tri_hit])
    assert (cpu["prim_id"] >= n_tris).sum() > 10_000
    assert (c_g["node_tests"], c_g["prim_tests"]) == (st_c["node_tests"], st_c["prim_tests"])
    assert np.array_equal(gsc.intersect_p(rays), osc.intersect_p(rays)[0])
    gsc.close()
    osc.close()


def test_gpu_hlbvh_adversarial_meshes(hip_ctx):
    """Hypothesis-generated small meshes (coincident vertices, zero-area triangles, repeated triangles, large and
    tiny coordinates): the device HLBVH equals the host builder byte for byte, and a scene over it answers rays
    like the oracle."""
    from hypothesis import given, settings, strategies as st
    from test_property_host import meshes

    @settings(max_examples=int(os.environ.get("PB_HYP_EXAMPLES", "150")), deadline=None)
    @given(mesh=meshes(), max_prims=st.sampled_from([1, 2, 4, 255]))
    def check(mesh, max_prims):
        verts, idx = mesh
        nodes_h, order_h = pbrt_hip.bvh_build(verts, idx, max_prims, pbrt_hip.SPLIT_HLBVH)
        nodes_g, order_g, _ = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, verts, idx, max_prims)
        assert np.array_equal(order_g, order_h) and nodes_g.tobytes() == nodes_h.tobytes()
        sc = dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
                  materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                  tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))
        rays = scenes.random_rays(256, 3, origin_extent=5.0)
        osc = oracle.OracleScene(sc, max_prims, 1)
        gsc = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=max_prims, device_build=True)
        try:
            cpu, _ = osc.intersect(rays)
            _assert_hits_equal(gsc.intersect(rays), cpu)
        finally:
            gsc.close()
            osc.close()

    check()


def test_too_deep_tree_is_refused(hip_ctx):
    """Trees deeper than the 64-entry traversal stack are refused at scene creation (no spill-slab overrun)."""
    n = 80
    x = (3.0 ** np.arange(n)).astype(np.float32)      # the midpoint of the centroid bounds peels off one triangle per level
    pos = np.zeros((3 * n, 3), dtype=np.float32)
    pos[0::3, 0], pos[1::3, 0], pos[2::3, 0] = x, x, x
    pos[1::3, 1], pos[2::3, 2] = 1e-3, 1e-3
    idx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    sc = dict(positions=pos, indices=idx, tri_material=np.zeros(n, dtype=np.int32),
              materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
              tri_light=np.full(n, -1, dtype=np.int32), lights=scenes._lights([]))
    with pytest.raises(pbrt_hip.PbrtHipError, match="deeper than the 64-entry"):
        pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=1, split_method=pbrt_hip.SPLIT_MIDDLE)
    g = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=1, split_method=pbrt_hip.SPLIT_SAH)   # a balanced tree is fine
    assert (g.intersect(scenes.random_rays(100, 1))["prim_id"] >= -1).all()
    g.close()


def test_error_behaviour_of_the_c_abi(hip_ctx):
    """Same error behaviour as the binding documents: bad arguments give an error code and a message, never a crash
    or a device fault (the reference would panic on most of these)."""
    sc = scenes.cornell_box()
    nodes, order = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH)

    def create(**changes):
        s2 = dict(sc)
        bvh = changes.pop("bvh", (nodes, order))
        s2.update(changes)
        return pbrt_hip.Scene(hip_ctx, s2, bvh=bvh)

    bad_idx = sc["indices"].copy()
    bad_idx[3, 1] = 10_000
    with pytest.raises(pbrt_hip.PbrtHipError, match="vertex index"):
        create(indices=bad_idx)
    bad_order = order.copy()
    bad_order[0] = bad_order[1]
    with pytest.raises(pbrt_hip.PbrtHipError, match="permutation"):
        create(bvh=(nodes, bad_order))
    bad_nodes = nodes.copy()
    bad_nodes["offset"][0] = 1                       # second child must come after the first subtree
    with pytest.raises(pbrt_hip.PbrtHipError, match="offset"):
        create(bvh=(bad_nodes, order))
    with pytest.raises(pbrt_hip.PbrtHipError, match="full binary tree|leaves do not cover|offset out of range"):
        create(bvh=(nodes[:-2].copy(), order))
    bad_mat = sc["tri_material"].copy()
    bad_mat[0] = 99
    with pytest.raises(pbrt_hip.PbrtHipError, match="tri_material"):
        create(tri_material=bad_mat)
    bad_materials = sc["materials"].copy()
    bad_materials["type"][0] = 17
    with pytest.raises(pbrt_hip.PbrtHipError, match="material type"):
        create(materials=bad_materials)
    bad_lights = sc["lights"].copy()
    bad_lights["prim"][0] = 10_000
    with pytest.raises(pbrt_hip.PbrtHipError, match="area light"):
        create(lights=bad_lights)
    bad_lights = sc["lights"].copy()
    bad_lights["type"][0] = 42
    with pytest.raises(pbrt_hip.PbrtHipError, match="light type"):
        create(lights=bad_lights)
    g = create()
    cam = scenes.cornell_camera(32, 32)
    for kw, msg in ((dict(integrator=9), "integrator"), (dict(sampler=("stratified", 0, 4, True, 4)), "stratified"),
                    (dict(bounds=(0, 0, 64, 64)), "bounds"), (dict(tile_rank=3, tile_world=2), "tile_rank"),
                    (dict(light_strategy=7), "light_strategy"), (dict(integrator=3, ao_samples=0), "ao_samples")):
        with pytest.raises(pbrt_hip.PbrtHipError, match=msg):
            g.render(cam, 32, 32, 2, **kw)
    bad_cam = cam.copy()
    bad_cam["kind"] = 5
    with pytest.raises(pbrt_hip.PbrtHipError, match="camera"):
        g.render(bad_cam, 32, 32, 2)
    film, st = g.render(cam, 32, 32, 2)               # the scene still works after all the refusals
    assert st["camera_samples"] == 32 * 32 * 2 and np.isfinite(film).all()
    # rays with NaN / infinite components are misses, not hangs
    rays = scenes.random_rays(64, 3)
    rays["d"][:16] = np.nan
    rays["o"][16:32] = np.inf
    rays["d"][32:48] = 0.0
    hits = g.intersect(rays)
    assert np.all(hits["prim_id"][:48] == -1)
    g.close()


def test_context_creation_from_several_threads_keeps_each_threads_error(hip_ctx):
    """pbrt_hip_context_create may be entered from several host threads at once; a creation that fails has no context to
    carry its message, so pbrt_hip_last_error(NULL) is the CALLING thread's (it was one unlocked global). Two
    threads keep failing with different reasons (device id out of range / a NULL out pointer leaves the text alone) while a
    third creates and destroys good contexts: every thread reads its own text, every time."""
    import ctypes
    import threading
    L = pbrt_hip.lib()
    L.pbrt_hip_last_error.restype = ctypes.c_char_p
    errors = []

    def bad(device_id):
        for _ in range(300):
            h = ctypes.c_void_p()
            rc = L.pbrt_hip_context_create(device_id, ctypes.byref(h))
            txt = L.pbrt_hip_last_error(None)
            if rc != 1 or txt != b"device_id out of range" or h.value:
                errors.append((device_id, rc, txt))

    def good():
        for _ in range(20):
            h = ctypes.c_void_p()
            rc = L.pbrt_hip_context_create(0, ctypes.byref(h))
            if rc != 0 or not h.value:
                errors.append(("good", rc, L.pbrt_hip_last_error(None)))
                continue
            L.pbrt_hip_context_destroy(h)
        if L.pbrt_hip_last_error(None) not in (b"", None):   # this thread never failed: its text is still empty
            errors.append(("good thread sees another thread's text", L.pbrt_hip_last_error(None)))

    threads = [threading.Thread(target=bad, args=(4096,)), threading.Thread(target=bad, args=(-1,)), threading.Thread(target=good)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:5]


def test_handles_shared_between_host_threads(hip_ctx):
    """The reference's Primitive is Sync + Send and li() is entered from many rayon tasks at once
    (src/core/primitive.rs:179, integrator.rs:412-452): several host threads call intersect / intersect_p / render on
    ONE scene (and a second scene of the same context) concurrently; every result equals the single-threaded one."""
    import threading
    sc_a = scenes.random_triangles(30_000, seq=4, size=0.03)
    sc_b = scenes.cornell_box()
    ga, gb = pbrt_hip.Scene(hip_ctx, sc_a), pbrt_hip.Scene(hip_ctx, sc_b)
    batches = [scenes.random_rays(20_000 + 1000 * k, 40 + k, origin_extent=1.5) for k in range(6)]
    want_hits = [ga.intersect(r) for r in batches]
    want_any = [ga.intersect_p(r) for r in batches]
    cam = scenes.cornell_camera(48, 32)
    want_film, _ = gb.render(cam, 48, 32, 4, max_depth=3, seed=9)
    errors, results = [], {}

    def worker(k):
        try:
            for rep in range(3):
                if k < 6:
                    results[("hit", k, rep)] = ga.intersect(batches[k])
                    results[("any", k, rep)] = ga.intersect_p(batches[k])
                else:
                    results[("film", k, rep)] = gb.render(cam, 48, 32, 4, max_depth=3, seed=9)[0]
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert len(results) == 6 * 3 * 2 + 2 * 3
    for (kind, k, rep), got in results.items():
        if kind == "hit":
            assert got.tobytes() == want_hits[k].tobytes()
        elif kind == "any":
            assert np.array_equal(got, want_any[k])
        else:
            assert np.array_equal(got, want_film)
    ga.close()
    gb.close()


def _aimed_rays(verts, idx, seed=7):
    """Rays aimed exactly at every vertex, edge midpoint and centroid of the mesh from origins at the target's own scale, a
    fifth of them with one direction component zeroed (as tests/test_gpu_wide.py's adversarial meshes)."""
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


def _scene_of(verts, idx):
    return dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
                materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))


@pytest.mark.parametrize("case", ["hypothesis counter-example", "the three triangles alone", "fan mesh"])
def test_t_max_that_moves_up_by_an_ulp(hip_ctx, case):
    """ray.t_max does not only shrink. Triangle::intersect_test accepts a hit when t_scaled <= t_max * det
    (triangle.rs:127-130) and then sets t = t_scaled / det, rounded once more: with two hits within an ulp of each other
    (a ray through a vertex three triangles share) the second t can come out ABOVE the t_max it was accepted under, and
    BVHAccel::intersect carries on with the larger value — a node popped later is tested against it (bvh.rs:841-842).
    Round 3's hypothesis run (3000 examples) found the mesh below: the binary kernel had dropped a far child at its parent
    because it lay behind the hit THEN. Fixed in both kernels: the binary one keeps a far child whenever the ray meets its
    slabs and decides at the pop, as the reference does; a ray whose t_max moves up leaves the wide kernel for the binary one."""
    verts = np.array([[0, 0, 1], [0, 0, -1e3], [0, 1, 1e3], [-1e3, 0.375, 0]], dtype=np.float32)
    if case == "hypothesis counter-example":
        idx = np.array([[0, 0, 0]] * 28 + [[0, 1, 3], [0, 2, 3], [1, 2, 3]], dtype=np.int32)
    elif case == "the three triangles alone":
        idx = np.array([[0, 1, 3], [0, 2, 3], [1, 2, 3]], dtype=np.int32)
    else:
        # closed fans around shared vertices at coordinates of a thousand units: many triangles meet in every vertex
        g = np.array([[x, y, 0.0] for y in range(7) for x in range(7)], dtype=np.float32)
        g[:, 2] = (np.sin(g[:, 0] * 1.3) + np.cos(g[:, 1] * 0.7)).astype(np.float32)
        verts = (g * np.float32(333.0) + np.float32(-1000.0)).astype(np.float32)
        quad = [(y * 7 + x, y * 7 + x + 1, (y + 1) * 7 + x + 1, (y + 1) * 7 + x) for y in range(6) for x in range(6)]
        idx = np.array([t for a, b, c, d in quad for t in ((a, b, c), (a, c, d))], dtype=np.int32)
    sc = _scene_of(verts, idx)
    rays = _aimed_rays(verts, idx)
    for max_prims, split in ((1, 0), (4, 0), (2, 2)):
        osc = oracle.OracleScene(sc, max_prims, split)
        gsc = pbrt_hip.Scene(hip_ctx, sc, max_prims_in_node=max_prims, split_method=split)
        cpu, _ = osc.intersect(rays)
        gpu = gsc.intersect(rays)
        for f in ("prim_id", "t", "b0", "b1", "b2"):
            bad = np.flatnonzero(gpu[f] != cpu[f])
            assert len(bad) == 0, (case, max_prims, split, f, bad[:5], gpu[bad[:3]], cpu[bad[:3]], gsc.wide_records())
        assert np.array_equal(gsc.intersect_p(rays), osc.intersect_p(rays)[0])
        gsc.close()
        osc.close()
