"""Property-based checks (hypothesis) of the host-side logic against the oracle: BVH builders on adversarial small
meshes (coincident vertices, zero-area triangles, collinear centroids, huge / tiny coordinates), the tile partition,
the sample bounds."""
import ctypes

import numpy as np
from hypothesis import given, settings, strategies as st

import oracle
import pbrt_hip

coord = st.one_of(st.sampled_from([0.0, 1.0, -1.0, 0.5, 1e-3, 1e3, -1e3, 3.0]),
                  st.floats(min_value=-4.0, max_value=4.0, allow_nan=False, width=32))


@st.composite
def meshes(draw):
    n_verts = draw(st.integers(min_value=3, max_value=24))
    verts = np.array(draw(st.lists(st.tuples(coord, coord, coord), min_size=n_verts, max_size=n_verts)), dtype=np.float32)
    verts = verts + np.float32(0.0)   # -0.0 -> +0.0: which zero min(+0, -0) returns is not defined by the reference (f32::min)
    n_tris = draw(st.integers(min_value=1, max_value=40))
    idx = np.array(draw(st.lists(st.tuples(*[st.integers(0, n_verts - 1)] * 3), min_size=n_tris, max_size=n_tris)), dtype=np.int32)
    return verts, idx


@settings(max_examples=120, deadline=None)
@given(mesh=meshes(), split=st.sampled_from([0, 1, 2, 3]), max_prims=st.sampled_from([1, 2, 4, 255]))
def test_host_builders_equal_oracle_on_adversarial_meshes(mesh, split, max_prims):
    verts, idx = mesh
    sc = dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
              materials=pbrt_hip.scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
              tri_light=np.full(len(idx), -1, dtype=np.int32), lights=pbrt_hip.scenes._lights([]))
    nodes, order = pbrt_hip.bvh_build(verts, idx, max_prims, split)
    osc = oracle.OracleScene(sc, max_prims, split)
    try:
        assert nodes.tobytes() == osc.nodes().tobytes()
        assert np.array_equal(order, osc.prim_order())
    finally:
        osc.close()
    assert sorted(order.tolist()) == list(range(len(idx)))
    leaf = nodes["n_primitives"] > 0
    assert int(nodes["n_primitives"].sum()) == len(idx) and leaf.sum() == (len(nodes) + 1) // 2


def _morton(tx, ty):
    code = 0
    for b in range(16):
        code |= ((tx >> b) & 1) << (2 * b) | ((ty >> b) & 1) << (2 * b + 1)
    return code


@settings(max_examples=200, deadline=None)
@given(x0=st.integers(-40, 40), y0=st.integers(-40, 40), dx=st.integers(0, 300), dy=st.integers(0, 200),
       world=st.integers(1, 9), order=st.sampled_from([pbrt_hip.TILE_ORDER_MORTON, pbrt_hip.TILE_ORDER_ROW_MAJOR]))
def test_tile_partition_is_a_partition(x0, y0, dx, dy, world, order):
    bounds = (x0, y0, x0 + dx, y0 + dy)
    seen = {}
    for rank in range(world):
        mine = [tuple(t) for t in pbrt_hip.tile_partition(bounds, rank, world, order).tolist()]
        for (tx, ty) in mine:
            assert (tx, ty) not in seen
            seen[(tx, ty)] = rank
            assert (tx - x0) % 16 == 0 and (ty - y0) % 16 == 0 and x0 <= tx < x0 + dx and y0 <= ty < y0 + dy
        # whatever the deal, a rank walks its own tiles in row-major order
        assert mine == sorted(mine, key=lambda t: (t[1], t[0]))
    n_tiles = ((dx + 15) // 16) * ((dy + 15) // 16)
    assert len(seen) == n_tiles
    # round-robin over the tiles in Morton order of the tile grid / in row-major order
    if order == pbrt_hip.TILE_ORDER_MORTON:
        dealt = sorted(seen, key=lambda t: _morton((t[0] - x0) // 16, (t[1] - y0) // 16))
    else:
        dealt = sorted(seen, key=lambda t: (t[1], t[0]))
    assert [seen[t] for t in dealt] == [i % world for i in range(n_tiles)]
    if order == pbrt_hip.TILE_ORDER_MORTON:   # pbrt_hip_tile_partition (no order argument) is the Morton deal
        for rank in range(world):
            n = ctypes.c_int32()
            assert pbrt_hip.lib().pbrt_hip_tile_partition(*bounds, rank, world, None, 0, ctypes.byref(n)) == 0
            out = np.zeros((n.value, 2), dtype=np.int32)
            assert pbrt_hip.lib().pbrt_hip_tile_partition(*bounds, rank, world, out.ctypes.data, n.value, ctypes.byref(n)) == 0
            assert out.tolist() == pbrt_hip.tile_partition(bounds, rank, world, order).tolist()


@settings(max_examples=200, deadline=None)
@given(w=st.integers(1, 4096), h=st.integers(1, 4096), rx=st.floats(0.5, 8.0, width=32), ry=st.floats(0.5, 8.0, width=32))
def test_sample_bounds_equal_oracle(w, h, rx, ry):
    assert pbrt_hip.sample_bounds(w, h, rx, ry) == oracle.sample_bounds(w, h, rx, ry)
