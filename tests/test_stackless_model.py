"""The stackless walk of csrc/trace_stackless.h as a model, on the CPU: parent links + one trail bit per level against the
reference's stack loop (src/accelerators/bvh.rs:828-932), over trees from the host builder (BVHAccel::new, no GPU needed).

What the kernel has to preserve is the ORDER of the reference's walk and the moment of every box test: near child first
(bvh.rs:857-865), the far child tested with the ray.t_max current when it is popped (bvh.rs:841-842). Both walks below record
the sequence of (leaf node, t_max at the visit); a leaf "hit" changes t_max by a rule that depends only on the ray and the leaf
— including, now and then, UPWARDS by a little, as the reference's rounded quotient can (triangle.rs:127-130) — so any
difference in order or in the t_max a box was tested with shows up in the sequences. The GPU kernel itself is checked bit for
bit against the oracle in tests/test_gpu_traversal_modes.py; this file pins the algorithm."""
import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes


def _slab(lo, hi, o, inv, neg, t_max):
    """Bounds3f::intersect_p with precomputed reciprocals (geometry.rs:709-751), float64: the order argument does not
    depend on the precision, only on both walks using the same test."""
    t0, t1 = 0.0, t_max
    for k in range(3):
        near, far = (hi[k], lo[k]) if neg[k] else (lo[k], hi[k])
        tn, tf = (near - o[k]) * inv[k], (far - o[k]) * inv[k]
        if tn > t0:
            t0 = tn
        if tf < t1:
            t1 = tf
        if t0 > t1:
            return False
    return True


def _leaf_event(ray_id, node, t_max):
    """What 'testing the leaf's triangles' does to t_max: a deterministic function of (ray, leaf). A third of the visits
    shorten the ray, one in sixteen lengthens it slightly (the ulp the reference's t can gain), the rest leave it."""
    h = (ray_id * 0x9E3779B1 + node * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 15
    if h % 3 == 0:
        return t_max * (0.4 + 0.5 * ((h >> 8) & 0xFF) / 255.0)
    if h % 16 == 1:
        return t_max * 1.0000001
    return t_max


def _walk_stack(nodes, o, d, ray_id, t_max):
    inv = 1.0 / d
    neg = inv < 0
    seq, stack, cur = [], [], 0
    while True:
        nd = nodes[cur]
        if _slab(nd["bmin"], nd["bmax"], o, inv, neg, t_max):
            if nd["n_primitives"] > 0:
                seq.append((cur, t_max))
                t_max = _leaf_event(ray_id, cur, t_max)
                if not stack:
                    break
                cur = stack.pop()
            elif neg[nd["axis"]]:
                stack.append(cur + 1)
                cur = int(nd["offset"])
            else:
                stack.append(int(nd["offset"]))
                cur = cur + 1
        else:
            if not stack:
                break
            cur = stack.pop()
    return seq


def _walk_stackless(nodes, parent, o, d, ray_id, t_max):
    """trace_stackless.h: going down, both children are tested, the near one is entered and a trail bit keeps whether the
    ray meets the far one's slabs at all; going up, a set bit means: test the far child NOW, with the current t_max."""
    inv = 1.0 / d
    neg = inv < 0
    seq = []
    root = nodes[0]
    if not _slab(root["bmin"], root["bmax"], o, inv, neg, t_max):
        return seq
    if root["n_primitives"] > 0:
        return [(0, t_max)]
    trail, cur, up = 0, 0, False          # cur: an interior node; up: it is being re-entered from below
    while cur >= 0:
        nd = nodes[cur]
        c = (cur + 1, int(nd["offset"]))
        near, far = (c[1], c[0]) if neg[nd["axis"]] else c
        hit = {k: _slab(nodes[k]["bmin"], nodes[k]["bmax"], o, inv, neg, t_max) for k in c}
        keep_far = _slab(nodes[far]["bmin"], nodes[far]["bmax"], o, inv, neg, np.inf)
        if not up and hit[near]:
            trail, child = (trail << 1) | int(keep_far), near
        elif (trail & 1 and hit[far]) if up else hit[far]:
            trail, child = (trail & ~1) if up else (trail << 1), far
        else:
            if up:
                trail >>= 1
            cur, up = parent[cur], True
            continue
        if nodes[child]["n_primitives"] > 0:
            seq.append((child, t_max))
            t_max = _leaf_event(ray_id, child, t_max)
            up = True                       # back into `cur` from below: its bit is bit 0 of the trail
        else:
            cur, up = child, False
        assert trail < (1 << 64)
    return seq


def _parents(nodes):
    parent = np.full(len(nodes), -1, dtype=np.int64)
    for i, nd in enumerate(nodes):
        if nd["n_primitives"] == 0:
            parent[i + 1] = parent[int(nd["offset"])] = i
    return parent


@pytest.mark.parametrize("n_tris,max_prims,split", [(1, 4, pbrt_hip.SPLIT_SAH), (2, 1, pbrt_hip.SPLIT_SAH), (300, 1, pbrt_hip.SPLIT_SAH),
                                                    (300, 4, pbrt_hip.SPLIT_HLBVH), (500, 2, pbrt_hip.SPLIT_MIDDLE),
                                                    (40, 1, pbrt_hip.SPLIT_EQUAL_COUNTS)])
def test_the_trail_walk_is_the_stack_walk(n_tris, max_prims, split):
    sc = scenes.random_triangles(n_tris, seq=n_tris + max_prims, size=0.2)
    nodes, _ = pbrt_hip.bvh_build(sc["positions"], sc["indices"], max_prims, split)
    parent = _parents(nodes)
    rays = scenes.random_rays(400, 3, origin_extent=1.2)
    visited = 0
    for i, r in enumerate(rays):
        o, d = r["o"].astype(np.float64), r["d"].astype(np.float64)
        if i % 7 == 0:
            d[i % 3] = 0.0                   # axis-parallel rays: infinite reciprocals
        if not d.any():
            d[2] = 1.0
        t_max = np.inf if i % 4 else 0.8
        with np.errstate(divide="ignore", invalid="ignore"):
            a = _walk_stack(nodes, o, d, i, t_max)
            b = _walk_stackless(nodes, parent, o, d, i, t_max)
        assert a == b, (i, a[:5], b[:5])
        visited += len(a)
    assert visited > 0


def test_a_chain_tree_uses_the_whole_trail():
    """SPLIT_MIDDLE over exponentially spaced triangles peels one off per level: 64 levels, 63 trail bits."""
    n = 64
    x = (3.0 ** np.arange(n)).astype(np.float32)
    pos = np.zeros((3 * n, 3), dtype=np.float32)
    pos[0::3, 0], pos[1::3, 0], pos[2::3, 0] = x, x, x
    pos[1::3, 1], pos[2::3, 2] = 1e-3, 1e-3
    nodes, _ = pbrt_hip.bvh_build(pos, np.arange(3 * n, dtype=np.int32).reshape(n, 3), 1, pbrt_hip.SPLIT_MIDDLE)
    parent = _parents(nodes)
    depth = np.zeros(len(nodes), dtype=np.int64)
    for i in range(1, len(nodes)):
        depth[i] = depth[parent[i]] + 1
    assert depth.max() == 63
    for i, (ox, dx) in enumerate([(-1.0, 1.0), (4e30, -1.0)]):
        o, d = np.array([ox, 2e-4, 2e-4]), np.array([dx, 0.0, 0.0])
        with np.errstate(divide="ignore", invalid="ignore"):
            a = _walk_stack(nodes, o, d, i, np.inf)
            b = _walk_stackless(nodes, parent, o, d, i, np.inf)
        assert a == b and len(a) == n
