"""What makes a flat tree a valid LinearBVHNode array (bvh.rs:95-104, 774-811), checked WITHOUT the oracle: depth-first order (an
interior node's first child is the next node, its second child at `offset`), every node reachable once, leaves that partition the
primitive order (each caller's triangle in exactly one leaf slot), leaf sizes within max_prims_in_node — except where a split fails
and the reference makes a bigger leaf (SAH's "leaf is cheaper", :440-470; identical centroids) —, every node's box = the union of
its children's = the box of the triangles below it, bit for bit, the split axis one of 0, 1, 2; under Middle and EqualCounts a leaf
of several primitives only where their centroids coincide. For the host builders of
libpbrt_hip.so (every split method) and, on a GPU, the device-built HLBVH."""
import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes
from test_brute_force import _height_field


def _validate(nodes, order, pos, idx, max_prims, name):
    n_tris = len(idx)
    assert sorted(order) == list(range(n_tris)), name                  # a permutation: each triangle in exactly one leaf slot
    tri = pos[idx]                                                     # Triangle::world_bound (triangle.rs:175-180)
    tmin, tmax = tri.min(axis=1), tri.max(axis=1)
    seen_nodes = np.zeros(len(nodes), dtype=np.int32)
    seen_slots = np.zeros(n_tris, dtype=np.int32)
    big_leaves = []

    def walk(i):
        seen_nodes[i] += 1
        nd = nodes[i]
        n = int(nd["n_primitives"])
        if n > 0:                                                      # leaf: `offset` is its first slot of the primitive order
            first = int(nd["offset"])
            assert 0 <= first and first + n <= n_tris, (name, i)
            seen_slots[first:first + n] += 1
            ids = order[first:first + n]
            if n > max_prims:
                big_leaves.append(ids)
            lo, hi = tmin[ids].min(axis=0), tmax[ids].max(axis=0)
        else:                                                          # interior: children at i + 1 and `offset`
            assert int(nd["axis"]) in (0, 1, 2), (name, i)
            second = int(nd["offset"])
            assert i + 1 < second < len(nodes), (name, i, second)
            lo0, hi0 = walk(i + 1)
            lo1, hi1 = walk(second)
            lo, hi = np.minimum(lo0, lo1), np.maximum(hi0, hi1)
        assert np.array_equal(nd["bmin"], lo) and np.array_equal(nd["bmax"], hi), (name, i)
        return lo, hi

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    walk(0)
    assert np.all(seen_nodes == 1) and np.all(seen_slots == 1), name   # every node and every slot reached exactly once
    return big_leaves


def _meshes():
    sc = scenes.random_triangles(4000, seq=9, extent=1.0, size=0.05)
    pos, idx = _height_field(20, 3)
    dup = np.repeat(scenes.random_triangles(50, seq=4, extent=1.0, size=0.1)["positions"].reshape(50, 3, 3), 9, axis=0)   # nine copies of each: centroids coincide
    return {"cloud": (sc["positions"], sc["indices"]), "height field": (pos, idx),
            "coinciding": (dup.reshape(-1, 3).astype(np.float32), np.arange(450 * 3, dtype=np.int32).reshape(450, 3)),
            "one": (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32), np.array([[0, 1, 2]], dtype=np.int32))}


@pytest.mark.parametrize("split", [pbrt_hip.SPLIT_SAH, pbrt_hip.SPLIT_HLBVH, pbrt_hip.SPLIT_MIDDLE, pbrt_hip.SPLIT_EQUAL_COUNTS])
@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_host_built_trees_are_valid(split, max_prims):
    for name, (pos, idx) in _meshes().items():
        nodes, order = pbrt_hip.bvh_build(pos, idx, max_prims, split)
        big = _validate(nodes, order, pos, idx, max_prims, f"{name} split {split} max_prims {max_prims}")
        if split in (pbrt_hip.SPLIT_MIDDLE, pbrt_hip.SPLIT_EQUAL_COUNTS):
            # these two split down to single primitives (max_prims_in_node is SAH's and HLBVH's business) unless the centroids — the
            # centres of the primitives' BOXES (bvh.rs:62: 0.5 min + 0.5 max) — coincide: the two triangles of a height-field cell
            tri = pos[idx]
            centroid = np.float32(0.5) * tri.min(axis=1) + np.float32(0.5) * tri.max(axis=1)
            leaves = nodes[nodes["n_primitives"] > 0]
            for l in leaves[leaves["n_primitives"] > 1]:
                c = centroid[order[l["offset"]:l["offset"] + l["n_primitives"]]]
                assert np.all(c == c[0]), (name, split)
            assert name != "cloud" or np.all(leaves["n_primitives"] == 1)
        n_leaves = int((nodes["n_primitives"] > 0).sum())
        assert len(nodes) == 2 * n_leaves - 1                          # a full binary tree


@pytest.mark.gpu
@pytest.mark.parametrize("max_prims", [1, 4])
def test_device_built_hlbvh_is_valid(hip_ctx, max_prims):
    for name, (pos, idx) in _meshes().items():
        nodes, order, _ = pbrt_hip.bvh_build_hlbvh_device(hip_ctx, pos, idx, max_prims)
        _validate(nodes, order, pos, idx, max_prims, f"{name} on the device, max_prims {max_prims}")
        assert len(nodes) == 2 * int((nodes["n_primitives"] > 0).sum()) - 1
