"""How many record steps per ray would a 4-wide layout chosen BY COST save over the fixed two-level collapse?

Host only. The oracle walks config 3's tree (the reference's loop, bvh.rs:828-932) for a set of rays — camera rays of the
frame and rays leaving random surface points in cosine-distributed directions (what the bounce wavefronts are), closest-hit
and any-hit — and counts, per binary node, the rays that pass its box test. A wide record is stepped when a ray reaches the
binary node it stands for, so steps(layout) = sum over the layout's record roots of visits[root]:
  fixed:     a record = a node and its two children (the shipped layout, wide_build.h: wide_slots_of)
  by cost:   a record = a node and up to two more interior nodes below it, chosen greedily by visit count among the interior
             children of what the record already holds (any of the five 4-leaf treelet shapes)
  optimal:   the tiling of the tree by such treelets that minimises the expected steps for THESE rays (dynamic programme over
             the eight ways a record can continue below its node): the most any cost function could get
The box tests behind `visits` are the exact ones (the 8-bit filter passes about 3 % more), the same for both layouts.
usage: python tools/wide_fill_study.py [n_triangles] [n_rays]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle, pbrt_hip
from pbrt_hip import scenes

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
sc = scenes.random_triangles(n_tris, seq=1)
osc = oracle.OracleScene(sc)
nodes = osc.nodes()
N = len(nodes)
rng = np.random.default_rng(1)

def ray_array(o, d, tmax):
    r = np.zeros((len(o), 8), dtype=np.float32)
    r[:, :3], r[:, 3:6], r[:, 6] = o, d, tmax
    return r

# camera rays of the 1920x1080 frame (random subset of pixels, jittered)
W, H = 1920, 1080
cam = scenes.camera_dict_to_floats(scenes.random_triangles_camera(W, H))
c2w, r2c = np.array(cam[:16]).reshape(4, 4), np.array(cam[16:32]).reshape(4, 4)
px = np.stack([rng.uniform(0, W, n_rays // 4), rng.uniform(0, H, n_rays // 4), np.zeros(n_rays // 4), np.ones(n_rays // 4)], 1)
pc = px @ r2c.T
pc = pc[:, :3] / pc[:, 3:4]
dc = pc / np.linalg.norm(pc, axis=1, keepdims=True)
cam_rays = ray_array(np.tile(c2w[:3, 3], (len(dc), 1)), dc @ c2w[:3, :3].T, np.inf)
# bounce rays: from random points on random triangles, cosine-distributed about the normal (either side)
pos, idx = sc["positions"], sc["indices"]
t = rng.integers(0, len(idx), n_rays)
b = rng.dirichlet([1, 1, 1], n_rays)
p = (pos[idx[t, 0]] * b[:, :1] + pos[idx[t, 1]] * b[:, 1:2] + pos[idx[t, 2]] * b[:, 2:3]).astype(np.float64)
n = np.cross(pos[idx[t, 1]] - pos[idx[t, 0]], pos[idx[t, 2]] - pos[idx[t, 0]]).astype(np.float64)
n /= np.linalg.norm(n, axis=1, keepdims=True)
n *= rng.choice([-1.0, 1.0], (n_rays, 1))
u1, u2 = rng.uniform(0, 1, n_rays), rng.uniform(0, 1, n_rays)
rr, ph = np.sqrt(u1), 2 * np.pi * u2
a = np.where(np.abs(n[:, :1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
s_ = np.cross(n, a); s_ /= np.linalg.norm(s_, axis=1, keepdims=True)
t_ = np.cross(n, s_)
d = s_ * (rr * np.cos(ph))[:, None] + t_ * (rr * np.sin(ph))[:, None] + n * np.sqrt(1 - u1)[:, None]
bounce = ray_array(p + n * 1e-4, d, np.inf)

L = oracle.lib()
def visits(rays, any_hit):
    v = np.zeros(N, dtype=np.uint64)
    r = np.ascontiguousarray(rays, dtype=np.float32)
    L.orc_node_visits(osc.h, r.ctypes.data_as(ctypes.c_void_p), len(r), int(any_hit), v.ctypes.data_as(ctypes.c_void_p))
    return v.astype(np.float64)

interior = nodes["n_primitives"] == 0
first = np.arange(N) + 1
second = nodes["offset"].astype(np.int64)

def record_roots_fixed():
    roots, work = [], [0]
    while work:
        i = work.pop()
        if not interior[i]:
            continue
        roots.append(i)
        for c in (first[i], second[i]):
            if interior[c]:
                work.extend([first[c], second[c]])   # grandchildren start the next records (leaf ones are skipped above)
    return np.array(roots)

def record_roots_by_cost(v):
    roots, work = [], [0]
    while work:
        i = work.pop()
        if not interior[i]:
            continue
        roots.append(i)
        slots = [first[i], second[i]]          # what the record holds so far: up to 4 children
        for _ in range(2):                     # absorb up to two more interior nodes, most visited first
            cand = [c for c in slots if interior[c]]
            if not cand or len(slots) >= 4:
                break
            c = max(cand, key=lambda k: v[k])
            slots.remove(c)
            slots += [first[c], second[c]]
        work.extend(slots)
    return np.array(roots)

def optimal_steps(v):
    """f(i) = visits[i] + min over the treelets rooted at i of the sum of f over the interior nodes just below the treelet."""
    f = np.zeros(N)
    fi, se, it = first.tolist(), second.tolist(), interior.tolist()
    vv = v.tolist()
    ff = [0.0] * N
    for i in range(N - 1, -1, -1):       # children have larger indices than their parent (depth-first order)
        if not it[i]:
            continue
        c0, c1 = fi[i], se[i]
        g = lambda k: ff[k] if it[k] else 0.0          # a child left outside the record starts its own (or is a leaf)
        best = g(c0) + g(c1)
        for a, b_ in ((c0, c1), (c1, c0)):
            if it[a]:
                al, ar = fi[a], se[a]
                inside = g(al) + g(ar)                 # a absorbed
                best = min(best, inside + g(b_))
                if it[b_]:
                    best = min(best, inside + g(fi[b_]) + g(se[b_]))       # both children absorbed (the fixed layout's shape)
                for x, y in ((al, ar), (ar, al)):
                    if it[x]:
                        best = min(best, g(fi[x]) + g(se[x]) + g(y) + g(b_))   # a and one of its children absorbed
        ff[i] = vv[i] + best
    return ff[0]

for name, rays, any_hit in (("camera rays, closest hit", cam_rays, 0), ("bounce rays, closest hit", bounce, 0), ("bounce rays, any hit", bounce, 1)):
    v = visits(rays, any_hit)
    fixed, cost = record_roots_fixed(), record_roots_by_cost(v)
    sf, scst, sopt = v[fixed].sum() / len(rays), v[cost].sum() / len(rays), optimal_steps(v) / len(rays)
    print(f"{name:26s} binary nodes entered per ray {v[interior].sum() / len(rays):6.1f} | records: fixed {len(fixed)} ({sf:5.1f} steps per ray), "
          f"by cost {len(cost)} ({scst:5.1f}: {100 * (1 - scst / sf):4.1f} % fewer steps, {100 * (1 - len(cost) / len(fixed)):4.1f} % fewer records), "
          f"optimal for these rays {sopt:5.1f} ({100 * (1 - sopt / sf):4.1f} % fewer steps)")
