"""Repro of the hypothesis counter-example of tests/test_gpu_wide.py::test_adversarial_meshes_... (round 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle, pbrt_hip
from pbrt_hip import scenes
verts = np.array([[0, 0, 1], [0, 0, -1e3], [0, 1, 1e3], [-1e3, 0.375, 0]], dtype=np.float32)
idx = np.array([[0, 0, 0]] * 28 + [[0, 1, 3], [0, 2, 3], [1, 2, 3]], dtype=np.int32)
max_prims, split = 1, 0
sc = dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
          materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
          tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))
tri = verts[idx]
targets = np.concatenate([tri.reshape(-1, 3), (tri[:, 0] + tri[:, 1]) * np.float32(0.5), tri.mean(axis=1)])
n = 3 * len(targets)
rays = scenes.random_rays(n, 7, origin_extent=2.0)
tgt = targets[np.arange(n) % len(targets)]
scale = np.maximum(np.abs(tgt).max(axis=1, keepdims=True), 1.0).astype(np.float32)
rays["o"] = (tgt + rays["o"] * scale).astype(np.float32)
rays["d"] = (tgt - rays["o"]).astype(np.float32)
k = np.arange(n); par = k % 5 == 0
rays["d"][par, k[par] % 3] = 0.0
rays["d"][np.all(rays["d"] == 0, axis=1)] = (0.0, 0.0, 1.0)
rays = np.ascontiguousarray(rays)
osc = oracle.OracleScene(sc, max_prims, split)
ctx = pbrt_hip.Context(0)
gsc = pbrt_hip.Scene(ctx, sc, max_prims_in_node=max_prims, split_method=split)
print("wide records:", gsc.wide_records(), "nodes", len(osc.nodes()))
cpu, _ = osc.intersect(rays)
gpu = gsc.intersect(rays)
ctx.set_counting(1); binr = gsc.intersect(rays); ctx.set_counting(0)   # the binary kernel (instrumented instantiation)
for name, got in (("wide", gpu), ("binary", binr)):
    bad = np.flatnonzero(~((got["prim_id"] == cpu["prim_id"]) & (got["t"] == cpu["t"]) & (got["b0"] == cpu["b0"]) & (got["b1"] == cpu["b1"]) & (got["b2"] == cpu["b2"])))
    print(name, "mismatches:", len(bad), "of", len(rays))
    for i in bad[:6]:
        print("  ray", i, "o", rays["o"][i], "d", rays["d"][i], "tmax", rays["tmax"][i] if "tmax" in rays.dtype.names else None)
        print("     oracle", cpu[i], "\n     gpu   ", got[i])
pw, pc = gsc.intersect_p(rays), osc.intersect_p(rays)[0]
print("any-hit mismatches:", int((pw != pc).sum()))
