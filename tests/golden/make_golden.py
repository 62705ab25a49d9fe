"""Generates the committed fixtures in tests/golden/ from the CPU oracle (oracle/liboracle.so).

The reference (lazytiger/pbrt-rs) can be neither built nor run and holds no fixtures for this path
(SURVEY.md §8c), so these vectors are produced by the oracle itself: they pin the oracle against
regressions and give the GPU tests inputs/outputs that travel to the GPU box. Data only.

  python tests/golden/make_golden.py      (rewrites the .npz / .json files next to this script)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
from pbrt_hip import scenes  # noqa: E402


def main():
    L = oracle.lib()
    # (iv) PCG32: first 16 outputs of sequences 0, 1, 2^32 (src/core/rng.rs)
    pcg = {}
    for seq in (0, 1, 2 ** 32):
        u = np.zeros(16, dtype=np.uint32)
        f = np.zeros(16, dtype=np.float32)
        L.orc_pcg32(seq, 16, u.ctypes.data, f.ctypes.data)
        pcg[str(seq)] = {"u32": [int(v) for v in u], "f32_hex": [float(v).hex() for v in f]}
    json.dump(pcg, open(os.path.join(HERE, "pcg32.json"), "w"), indent=1)

    # (i) triangle table: 64 seeded ray / triangle pairs -> (hit, b0, b1, b2, t)
    n = 64
    u = scenes.pcg32_float(77, n * 15).reshape(n, 15)
    tris = (u[:, :9] * 2 - 1).astype(np.float32).reshape(n, 3, 3)
    o = (u[:, 9:12] * 4 - 2).astype(np.float32)
    target = tris.mean(axis=1) + (u[:, 12:15] - 0.5).astype(np.float32) * np.float32(0.8)
    d = (target - o).astype(np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6] = o, d, np.inf
    rays[::5, 6] = 0.9
    out = np.zeros((n, 5), dtype=np.float32)
    for i in range(n):
        p = np.ascontiguousarray(tris[i])
        L.orc_triangle_test(p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data, rays[i].ctypes.data, 0,
                            out[i].ctypes.data)
    np.savez(os.path.join(HERE, "triangle_kat.npz"), tris=tris, rays=rays, out=out)

    # (v) 4096-ray hit table on a seeded 1k-triangle scene, with the reference-loop visit counts
    sc = scenes.random_triangles(1000, seq=42, size=0.15)
    osc = oracle.OracleScene(sc)
    r = scenes.random_rays(4096, 43, origin_extent=1.5)
    r["t_max"][::7] = np.float32(1.25)
    hits, ctr = osc.intersect(r, n_threads=1)
    occl, ctr_p = osc.intersect_p(r, n_threads=1)
    np.savez(os.path.join(HERE, "hit_table_1k.npz"), rays=r, hits=hits, occluded=occl,
             counters=np.array([ctr["rays"], ctr["node_tests"], ctr["prim_tests"], ctr_p["rays"],
                                ctr_p["node_tests"], ctr_p["prim_tests"]], dtype=np.uint64),
             nodes=osc.nodes(), prim_order=osc.prim_order())
    osc.close()

    # (vi) 64x64x4spp Cornell (path, depth 8) and mixed-material images, raw f32 film {xyz, weight}
    w = h = 64
    osc = oracle.OracleScene(scenes.cornell_box())
    film, st = osc.render(scenes.camera_dict_to_floats(scenes.cornell_camera(w, h)), w, h, 4, max_depth=8, seed=0,
                          n_threads=1)
    np.savez_compressed(os.path.join(HERE, "cornell_64x64x4.npz"), film=film,
                        stats=np.array([st["rays"], st["node_tests"], st["prim_tests"], st["camera_samples"]],
                                       dtype=np.uint64))
    osc.close()
    osc = oracle.OracleScene(scenes.mixed_materials_scene())
    film, st = osc.render(scenes.camera_dict_to_floats(scenes.random_triangles_camera(w, h)), w, h, 4, max_depth=16,
                          seed=5, n_threads=1)
    np.savez_compressed(os.path.join(HERE, "mixed_64x64x4.npz"), film=film,
                        stats=np.array([st["rays"], st["node_tests"], st["prim_tests"], st["camera_samples"]],
                                       dtype=np.uint64))
    osc.close()
    # (vii) one small render per widened row (variants.py): films + ray counts
    from variants import W, H, SPP, oracle_scene, variants
    out = {}
    for name, (sc, cam, kw) in variants().items():
        kw = dict(kw)
        spec = kw.pop("filter_spec", None)
        if spec is not None:
            kw["filter"] = (spec[1], spec[1], oracle.filter_table(spec[0], spec[1], spec[1], spec[2], spec[3]))
        osc = oracle_scene(oracle, sc)
        film, st = osc.render(scenes.camera_dict_to_floats(cam), W, H, SPP, n_threads=1, **kw)
        osc.close()
        out[name] = film
        out[name + "__rays"] = np.array([st["rays"], st["camera_samples"]], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "variants_48x32.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
