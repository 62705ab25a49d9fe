"""Probe: does ray order matter to the traversal kernel? Bounce-like rays (origins on surfaces of the config-3
cloud, uniform directions) traced in random order, sorted by the Morton code of the origin, and sorted by origin +
direction octant. Trace-kernel time only (HIP events inside the library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np, pbrt_hip
from pbrt_hip import scenes

n = int(os.environ.get("RAYS", "8000000"))
sc = scenes.random_triangles(1_000_000, seq=1)
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc)
first = scenes.random_rays(2 * n, 77, origin_extent=1.2)
hits = g.intersect(first)
ok = np.nonzero(hits["prim_id"] >= 0)[0][:n]
rays = np.zeros(len(ok), dtype=scenes.RAY_DTYPE)
rays["o"] = first["o"][ok] + hits["t"][ok, None] * first["d"][ok] * np.float32(0.999)
d = scenes.pcg32_float(9, len(ok) * 3).reshape(-1, 3) * 2 - 1
rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
rays["t_max"] = np.inf


def morton(p, bits=10):
    q = np.clip(((p + 1.3) / 2.6 * (1 << bits)).astype(np.uint32), 0, (1 << bits) - 1)
    code = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + k)
    return code


def timed(label, r):
    g.intersect(r[:1000])
    ctx.trace_timing(reset=True)
    out = g.intersect(r)
    ms, launches = ctx.trace_timing(reset=True)
    print(f"{label:40s} {len(r) / ms * 1e-3:8.1f} Mrays/s  ({ms:.2f} ms)", flush=True)
    return out


timed("random order", rays)
code = morton(rays["o"])
timed("sorted by origin Morton code", rays[np.argsort(code, kind="stable")])
octant = ((rays["d"][:, 0] < 0).astype(np.uint64) | ((rays["d"][:, 1] < 0).astype(np.uint64) << np.uint64(1)) |
          ((rays["d"][:, 2] < 0).astype(np.uint64) << np.uint64(2)))
timed("sorted by direction octant, then origin", rays[np.argsort((octant << np.uint64(30)) | code, kind="stable")])
timed("sorted by origin (5 bits/axis), then octant", rays[np.argsort(((morton(rays["o"], 5)) << np.uint64(3)) | octant, kind="stable")])
