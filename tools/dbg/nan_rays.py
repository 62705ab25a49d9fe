"""Non-finite rays (NaN / inf in origin, direction, t_max): does the GPU answer what the oracle answers?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle, pbrt_hip, fuzz_intersect as F
from pbrt_hip import scenes
ctx = pbrt_hip.Context(0)
tot = bad = 0
for seed in range(300):
    sc, rays, max_prims, split, kw, n_inst, desc = F.make_case(seed)
    rng = np.random.default_rng(seed + 99)
    k = rng.choice(len(rays), min(len(rays), 200), replace=False)
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e38, -1e38, 1e-45], dtype=np.float32)
    for j, i in enumerate(k):
        f = ("o", "d", "t_max")[j % 3]
        if f == "t_max":
            rays["t_max"][i] = vals[rng.integers(0, len(vals))]
        else:
            rays[f][i, rng.integers(0, 3)] = vals[rng.integers(0, len(vals))]
    rays = np.ascontiguousarray(rays[k])
    osc = oracle.OracleScene(sc, max_prims, split); gsc = pbrt_hip.Scene(ctx, sc, max_prims_in_node=max_prims, split_method=split, **kw)
    cpu, _ = osc.intersect(rays); occ = osc.intersect_p(rays)[0]
    g = gsc.intersect(rays); p = gsc.intersect_p(rays)
    gsc.close(); osc.close()
    m = (g["prim_id"] != cpu["prim_id"]) | (p != occ)
    hit = cpu["prim_id"] >= 0
    for f in ("t", "b0", "b1", "b2"):
        m |= hit & ~((g[f] == cpu[f]) | (np.isnan(g[f]) & np.isnan(cpu[f])))
    tot += len(rays); bad += int(m.sum())
    if m.any() and bad < 40:
        i = int(np.flatnonzero(m)[0])
        print(desc, "\n  ray", rays[i], "\n  oracle", cpu[i], occ[i], "\n  gpu   ", g[i], p[i])
print("non-finite rays:", tot, "mismatching", bad)
