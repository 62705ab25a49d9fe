"""Host-side proof obligations of the 4-wide quantised records (pbrt-rs_amd/csrc/wide_bvh.h, host_wide.cpp), no GPU.

The traversal kernel over those records returns the reference's results bit for bit provided (wide_bvh.h):
  P1  the 8-bit box filter never rejects what Bounds3f::intersect_p (src/core/geometry.rs:709-751) accepts for a leaf
      box below it, and its entry distance never exceeds the exact one;
  P2  ranking a record's children by the two dir_is_neg[axis] levels gives BVHAccel::intersect's visiting order
      (src/accelerators/bvh.rs:857-865);
  P3  every leaf is referenced once, with its triangles and its exact box.
tests/native/wide_check.cpp compiles the product's builder and the product's filter arithmetic (the very functions the
kernel calls) for the host and checks the three on real trees with adversarial rays; its restatement of the slab test
is pinned against the oracle's here. The GPU parity tests (test_gpu_intersect.py, test_gpu_render.py) then check the
kernel end to end.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "wide_check.cpp")
OUT = os.path.join(HERE, "native", "_build", "libwide_check.so")


@pytest.fixture(scope="module")
def chk():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(HERE, "..", "pbrt-rs_amd", "csrc", f) for f in ("host_wide.cpp", "host_wide.h", "wide_bvh.h", "wide_build.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", OUT, SRC])
    L = ctypes.CDLL(OUT)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.wide_check_structure.argtypes = [vp, i32, vp, i32, vp, ctypes.c_char_p, ctypes.c_int]
    L.wide_check_filter.argtypes = [vp, i32, vp, i32, vp, i32, vp]
    L.wide_check_filter.restype = ctypes.c_int64
    L.wide_check_slab.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp]
    return L


def _tri_records(positions, indices, order):
    """The scene's 48-B leaf-order triangle records (pbrt_hip.hip scene_create_impl): 9 vertex floats, prim, material, flags."""
    tri = positions[indices[order]].reshape(-1, 9).astype(np.float32)
    rec = np.zeros((len(order), 12), dtype=np.float32)
    rec[:, :9] = tri
    rec[:, 9] = order.astype(np.int32).view(np.float32)
    return np.ascontiguousarray(rec)


def _tree(sc, max_prims=4, split=pbrt_hip.SPLIT_SAH):
    pos = np.ascontiguousarray(sc["positions"], dtype=np.float32)
    idx = np.ascontiguousarray(sc["indices"], dtype=np.int32)
    nodes, order = pbrt_hip.bvh_build(pos, idx, max_prims, split)
    return nodes, _tri_records(pos, idx, order)


def _structure(chk, nodes, tris):
    stats = np.zeros(4, dtype=np.int64)
    why = ctypes.create_string_buffer(200)
    rc = chk.wide_check_structure(nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris), stats.ctypes.data, why, 200)
    return rc, stats, why.value.decode()


SCENES = {
    "cornell": lambda: scenes.cornell_box(),
    "rand20k": lambda: scenes.random_triangles(20_000, seq=3, size=0.05),
    "rand3k_big": lambda: scenes.random_triangles(3_000, seq=5, size=0.4),
    "mixed": lambda: scenes.mixed_materials_scene(),
}


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("max_prims,split", [(4, 0), (1, 0), (2, 1), (4, 2), (3, 3)])
def test_structure_and_order(chk, name, max_prims, split):
    nodes, tris = _tree(SCENES[name](), max_prims, split)
    rc, stats, why = _structure(chk, nodes, tris)
    assert rc == 0, (rc, why)
    n_leaves = int((nodes["n_primitives"] > 0).sum())
    assert stats[2] == n_leaves
    if len(nodes) > 1:
        assert 0 < stats[0] <= (len(nodes) - 1) // 2   # at most one record per interior node, usually about half
        assert stats[1] <= 3 * 64


def test_builder_declines_what_the_argument_does_not_cover(chk):
    sc = scenes.random_triangles(500, seq=9, size=0.2)
    nodes, tris = _tree(sc, 255)   # leaves of up to 255 primitives
    if nodes["n_primitives"].max() > 4:
        rc, _, why = _structure(chk, nodes, tris)
        assert rc == -100 and "more than 4" in why
    nodes, tris = _tree(sc, 4)
    far = nodes.copy()
    far["bmax"][0, 0] = 3e6   # coordinates beyond 2^20
    assert _structure(chk, far, tris)[0] == -100
    loose = nodes.copy()
    leaf1 = np.flatnonzero(nodes["n_primitives"] == 1)[0]
    loose["bmin"][leaf1] -= np.float32(0.5)   # a single-triangle leaf whose box is not the triangle's bounds
    rc, _, why = _structure(chk, loose, tris)
    assert rc == -100 and ("not the triangle" in why or "not inside" in why)


@pytest.mark.parametrize("offset,declined", [(0.0, False), (60.0, False), (60000.0, True)])
def test_builder_declines_scenes_the_planes_would_not_filter(chk, offset, declined):
    """The descriptor bytes ride in the low mantissa bytes of base.xyz, which moves the base down by up to 256 ulp of the
    COORDINATE: around 6e4 that is a whole unit, the cells of a node 0.05 across come out at 1 / 255 and its 8-bit planes
    filter nothing. Hits would stay exact; steps per ray would not. Such a tree keeps the binary records."""
    sc = scenes.random_triangles(30_000, seq=6, size=0.03)
    sc = dict(sc, positions=(sc["positions"] + np.float32(offset)).astype(np.float32))
    nodes, tris = _tree(sc, 4)
    rc, _, why = _structure(chk, nodes, tris)
    if declined:
        assert rc == -100 and "too far from the origin" in why, (rc, why)
    else:
        assert rc == 0, (rc, why)


def test_slab_restatement_is_the_oracles(chk):
    """wide_check.cpp's slab test == the oracle's Bounds3f::intersect_p on random and degenerate cases."""
    L = oracle.lib()
    g = scenes.pcg32_float(77, 6000 * 13).reshape(6000, 13)
    n_pass = 0
    for k in range(6000):
        lo = (g[k, 0:3] * 2 - 1).astype(np.float32)
        hi = lo + (g[k, 3:6] * (0.0 if k % 7 == 0 else 0.5)).astype(np.float32)
        o = (g[k, 6:9] * 4 - 2).astype(np.float32)
        d = (g[k, 9:12] * 2 - 1).astype(np.float32)
        if k % 2 == 0:                    # aimed near the box, so that both outcomes are frequent
            d = ((lo + hi) * np.float32(0.5) - o + d * np.float32(0.3)).astype(np.float32)
        if k % 5 == 0:
            o[k % 3] = lo[k % 3]          # origin exactly on a face
        if k % 11 == 0:
            d[(k // 11) % 3] = 0.0        # axis-parallel: infinite reciprocal
        tmax = np.float32(np.inf if k % 3 else g[k, 12] * 3)
        with np.errstate(divide="ignore"):
            inv = (np.float32(1) / d).astype(np.float32)
        box = np.concatenate([lo, hi]).astype(np.float32)
        ray = np.array(list(o) + list(d) + [tmax, 0], dtype=np.float32)
        entry = ctypes.c_float()
        mine = chk.wide_check_slab(lo.ctypes.data, hi.ctypes.data, o.ctypes.data, inv.ctypes.data, tmax, ctypes.byref(entry))
        assert mine == L.orc_bounds_intersect_p(box.ctypes.data, ray.ctypes.data, 0), k
        n_pass += mine
    assert 300 < n_pass < 5700


def _adversarial_rays(nodes, n, seq):
    """Random rays plus the cases rounding bites on: origins exactly on box planes, tiny direction components (huge
    reciprocals, still inside the covered range), distant origins, grazing directions along box faces."""
    u = scenes.pcg32_float(seq, n * 8).reshape(n, 8)
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    ext = hi - lo
    o = lo + (u[:, :3] * 1.4 - 0.2) * ext
    d = u[:, 3:6] * 2 - 1
    leaves = np.flatnonzero(nodes["n_primitives"] > 0)
    pick = leaves[(u[:, 6] * len(leaves)).astype(np.int64) % len(leaves)]
    k = np.arange(n)
    a = k % 3
    on_face = k % 4 == 0            # origin coordinate exactly on a leaf box plane
    o[on_face, a[on_face]] = np.where(k[on_face] % 8 == 0, nodes["bmin"][pick[on_face], a[on_face]],
                                      nodes["bmax"][pick[on_face], a[on_face]])
    tiny = k % 5 == 1               # one direction component ~1e-9 .. 1e-11: reciprocal up to 1e11 < 2^40
    d[tiny, a[tiny]] = np.where(k[tiny] % 2 == 0, 1e-9, -3e-11)
    far = k % 7 == 2                # origin far outside, aimed at a leaf
    o[far] = lo + ext * 0.5 + (u[far, :3] - 0.5) * 4000.0
    centre = (nodes["bmin"][pick] + nodes["bmax"][pick]) * 0.5
    aim = (k % 7 == 2) | (k % 3 == 0)
    d[aim] = centre[aim] - o[aim] + (u[aim, 3:6] - 0.5) * 0.02 * ext
    graze = k % 9 == 4              # towards a corner of a leaf box: the silhouette cases
    d[graze] = nodes["bmax"][pick[graze]] - o[graze]
    d[np.all(d == 0, axis=1)] = (0.3, 0.2, 0.1)
    rays = np.zeros((n, 7), dtype=np.float32)
    rays[:, :3] = o
    rays[:, 3:6] = d
    rays[:, 6] = np.where(k % 2 == 0, np.inf, u[:, 7] * 2.0 * np.linalg.norm(ext) / np.maximum(np.linalg.norm(d, axis=1), 1e-6))
    return np.ascontiguousarray(rays)


@pytest.mark.parametrize("name,n_rays", [("cornell", 6000), ("rand20k", 1500), ("rand3k_big", 3000), ("mixed", 4000)])
def test_filter_is_conservative(chk, name, n_rays):
    nodes, tris = _tree(SCENES[name]())
    rays = _adversarial_rays(nodes, n_rays, 101)
    counts = np.zeros(4, dtype=np.int64)
    bad = chk.wide_check_filter(nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris), rays.ctypes.data, len(rays),
                                counts.ctypes.data)
    assert bad == 0, (bad, counts)
    assert counts[0] > n_rays * 0.9            # nearly all rays are inside the covered range
    assert counts[1] > n_rays                  # and the property was exercised
    # the pad is per axis: a ray nearly parallel to a slab (huge plane values on that axis) must not open the filter
    # on the other two (with one pad for all axes such rays walked the whole tree: a 10 ms tail per launch on the GPU)
    assert counts[3] <= counts[2] <= counts[3] * 1.3 + 50, counts
    # ... and it is not vacuous: for ordinary rays it passes only a few percent more leaf children than the exact test
    r = scenes.random_rays(400, 5, origin_extent=float(np.abs(nodes["bmax"][0]).max()))
    plain = np.zeros((400, 7), dtype=np.float32)
    plain[:, :3], plain[:, 3:6], plain[:, 6] = r["o"], r["d"], np.inf
    c2 = np.zeros(4, dtype=np.int64)
    assert chk.wide_check_filter(nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris), plain.ctypes.data, 400, c2.ctypes.data) == 0
    assert c2[3] <= c2[2] <= c2[3] * 1.15 + 20, c2


def test_filter_on_an_axis_aligned_scene_with_rays_on_its_planes(chk):
    """Cornell box: every box plane is a wall plane; rays start ON walls and run along them."""
    sc = scenes.cornell_box()
    nodes, tris = _tree(sc)
    n = 4000
    u = scenes.pcg32_float(55, n * 6).reshape(n, 6)
    rays = np.zeros((n, 7), dtype=np.float32)
    rays[:, :3] = u[:, :3] * 555.0
    rays[:, 3:6] = u[:, 3:] * 2 - 1
    k = np.arange(n)
    walls = np.array([0.0, 555.0, 548.8, 559.2, 165.0, 330.0], dtype=np.float32)
    rays[k, k % 3] = walls[k % 6]                 # a coordinate exactly on a wall / block plane
    along = k % 4 == 0
    rays[along, 3 + (k[along] % 3)] = 1e-10       # and nearly parallel to it
    rays[:, 6] = np.inf
    counts = np.zeros(4, dtype=np.int64)
    bad = chk.wide_check_filter(nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris), rays.ctypes.data, n, counts.ctypes.data)
    assert bad == 0 and counts[1] > n


def test_top_level_tree_over_opaque_primitives(chk):
    """A two-level scene's top-level aggregate (src/core/api.rs:1481-1520: TransformedPrimitives beside world triangles):
    the builder gets no triangle records (tris == NULL), keeps every leaf's exact box and reports the wide-order ->
    leaf-slot map; same P1-P4."""
    sc = scenes.two_level_scene(n_instances=60)
    trees, inst, tlas_nodes, tlas_order = pbrt_hip.build_general_two_level(sc)
    n_slots = len(tlas_order)
    stats = np.zeros(4, dtype=np.int64)
    why = ctypes.create_string_buffer(200)
    rc = chk.wide_check_structure(tlas_nodes.ctypes.data, len(tlas_nodes), None, n_slots, stats.ctypes.data, why, 200)
    assert rc == 0, (rc, why.value)
    assert stats[2] == int((tlas_nodes["n_primitives"] > 0).sum())
    rays = _adversarial_rays(tlas_nodes, 3000, 17)
    counts = np.zeros(4, dtype=np.int64)
    bad = chk.wide_check_filter(tlas_nodes.ctypes.data, len(tlas_nodes), None, n_slots, rays.ctypes.data, len(rays), counts.ctypes.data)
    assert bad == 0 and counts[1] > 1000, (bad, counts)
    # ... and each object aggregate is an ordinary triangle tree
    for o, (nodes, order) in zip(sc["objects"], trees):
        tris = _tri_records(np.asarray(o["positions"], dtype=np.float32), np.asarray(o["indices"], dtype=np.int32), order)
        rc, st, w = _structure(chk, nodes, tris)
        assert rc == 0, (rc, w)


def test_record32_study_tool_runs_and_agrees_with_itself(tmp_path):
    """tools/record32_study.py (profiles/r05_record32_study.txt: the 32-byte two-load record costed on the host, not built): on a
    small tree both layouts are walked by the same rays; they must find the same hits (equal triangle-test counts up to the few
    extra leaf candidates of a looser filter) and the candidate must save a third of the record loads — the tool still says what
    the profile says it says."""
    import re
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "record32_study.py"), "20000", "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    tail = r.stdout[r.stdout.index("all rays (the frame's mix)"):]
    steps48, steps32 = float(re.search(r"48-B records:\s+([\d.]+) steps/ray", tail).group(1)), float(re.search(r"32-B records:\s+([\d.]+) steps/ray", tail).group(1))
    tri48, tri32 = [float(v) for v in re.findall(r"([\d.]+) triangle tests", tail)[:2]]
    assert 0.99 * steps48 <= steps32 <= 1.03 * steps48 and abs(tri48 - tri32) <= 0.02 * tri48
    req = re.search(r"requests per ray:\s+([\d.]+) ->\s+([\d.]+)", tail)
    assert 0.66 <= float(req.group(2)) / float(req.group(1)) <= 0.78   # between "every load is a record load" (2 / 3) and the full frame's 0.73
    assert "frame origins that would not be exact floats: 0" in r.stdout
