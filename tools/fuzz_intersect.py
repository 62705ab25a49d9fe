"""Fuzz campaign for Scene::intersect / intersect_p: random meshes WITH shared vertices and edges (height fields, fans, duplicated
triangles) at several coordinate scales, rays aimed exactly at vertices, edge midpoints and centroids (where equal-t ties and
one-ulp differences live) plus random rays; the wide kernel (default), the binary kernel (instrumented instantiation) and the
oracle must agree bit for bit on (prim, t, b0, b1, b2) and on occlusion.
usage: python tools/fuzz_intersect.py [n_cases] [first_seed]   (GPU box; exits 1 if any case mismatches)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle, pbrt_hip
from pbrt_hip import scenes


BIG = bool(os.environ.get("FUZZ_BIG"))   # FUZZ_BIG=1: meshes of 10^4 - 10^5 triangles (deep trees, stack spills), 10^5 rays per case


def make_mesh(rng):
    scale = float(rng.choice([1e-3, 1.0, 1.0, 37.0, 1000.0]))
    parts_p, parts_i, base = [], [], 0
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["grid", "fan", "cloud", "dup"])
        if kind == "grid":
            n = int(rng.integers(40, 160)) if BIG else int(rng.integers(2, 12))
            xs = np.linspace(-1, 1, n)
            amp = float(rng.choice([0.0, 0.05, 0.5]))
            p = np.array([[x, amp * np.sin(3 * x + 2 * z), z] for z in xs for x in xs])
            q = [(z * n + x, z * n + x + 1, (z + 1) * n + x + 1, (z + 1) * n + x) for z in range(n - 1) for x in range(n - 1)]
            i = np.array([t for a, b, c, d in q for t in ((a, b, c), (a, c, d))])
        elif kind == "fan":
            k = int(rng.integers(3, 40))
            ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
            p = np.concatenate([[[0.0, float(rng.uniform(-0.5, 0.5)), 0.0]], np.stack([np.cos(ang), rng.uniform(-0.2, 0.2, k), np.sin(ang)], 1)])
            i = np.array([(0, 1 + j, 1 + (j + 1) % k) for j in range(k)])
        elif kind == "cloud":
            n = int(rng.integers(10_000, 150_000)) if BIG else int(rng.integers(1, 300))
            c = scenes.random_triangles(n, seq=int(rng.integers(1, 10000)), extent=1.0, size=float(rng.choice([0.01, 0.05]) if BIG else rng.choice([0.02, 0.2])))
            p, i = c["positions"].astype(np.float64), c["indices"]
        else:   # the same triangle several times under different primitive numbers, and a degenerate one
            p = rng.uniform(-1, 1, (3, 3))
            i = np.array([(0, 1, 2)] * int(rng.integers(2, 6)) + [(0, 0, 1)])
        off = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 0.5]))
        parts_p.append((p + off) * scale)
        parts_i.append(np.asarray(i) + base)
        base += len(p)
    return np.concatenate(parts_p).astype(np.float32), np.concatenate(parts_i).astype(np.int32), scale


def aimed_rays(rng, verts, idx, scale, n_random):
    tri = verts[idx]
    targets = np.concatenate([tri.reshape(-1, 3), (tri[:, 0] + tri[:, 1]) * np.float32(0.5), (tri[:, 1] + tri[:, 2]) * np.float32(0.5), tri.mean(axis=1)])
    cap = 15_000 if BIG else 600
    if len(targets) > cap:
        targets = targets[rng.choice(len(targets), cap, replace=False)]
    reps = 3
    tgt = np.repeat(targets, reps, axis=0)
    o = tgt + rng.normal(size=tgt.shape) * scale * rng.choice([0.5, 3.0, 30.0], (len(tgt), 1))
    rays = np.zeros(len(tgt) + n_random, dtype=pbrt_hip.RAY_DTYPE)
    rays["o"][:len(tgt)] = o.astype(np.float32)
    rays["d"][:len(tgt)] = (tgt - rays["o"][:len(tgt)]).astype(np.float32)
    k = np.arange(len(tgt))
    par = k % 7 == 0
    rays["d"][:len(tgt)][par, k[par] % 3] = 0.0                        # axis-parallel among them
    rr = scenes.random_rays(n_random, int(rng.integers(1, 1 << 20)), origin_extent=2.0)
    rays["o"][len(tgt):] = rr["o"] * np.float32(scale)
    rays["d"][len(tgt):] = rr["d"]
    rays["t_max"] = np.inf
    some = rng.random(len(rays)) < 0.2
    rays["t_max"][some] = (rng.uniform(0.2, 1.5, some.sum())).astype(np.float32)  # finite t_max: around the aimed hits' t = 1
    rays["d"][np.all(rays["d"] == 0, axis=1)] = (0.0, 0.0, 1.0)
    # one ray in a hundred with a non-finite, denormal or huge component: whatever the reference's comparisons make of it
    odd = np.flatnonzero(rng.random(len(rays)) < 0.01)
    vals = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e38, -1e38, 1e-45], dtype=np.float32)
    for j, i in enumerate(odd):
        f = ("o", "d", "t_max")[j % 3]
        if f == "t_max":
            rays["t_max"][i] = vals[rng.integers(0, len(vals))]
        else:
            rays[f][i, rng.integers(0, 3)] = vals[rng.integers(0, len(vals))]
    return np.ascontiguousarray(rays)


def make_case(seed):
    rng = np.random.default_rng(seed)
    verts, idx, scale = make_mesh(rng)
    sc = dict(positions=verts, indices=idx, tri_material=np.zeros(len(idx), dtype=np.int32),
              materials=scenes._materials([(1, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
              tri_light=np.full(len(idx), -1, dtype=np.int32), lights=scenes._lights([]))
    n_inst = 0
    kw = {}
    r_kind = rng.random()
    if r_kind < 0.15:
        # the general two-level scene (primitive.rs:105-159): several object aggregates, instances of them, world-space
        # triangles beside the instances in the top-level leaves
        objs, n_obj = [], int(rng.integers(1, 4))
        for _ in range(n_obj):
            ov, oi, _s = make_mesh(rng)
            objs.append(dict(positions=(ov / np.float32(_s) * np.float32(scale)).astype(np.float32), indices=oi, tri_material=np.zeros(len(oi), dtype=np.int32)))
        n_inst = int(rng.integers(1, 20))
        inst = np.zeros((n_inst, 2, 4, 4), dtype=np.float32)
        io = rng.integers(0, n_obj, n_inst).astype(np.int32)
        tgt_sets = []
        for k in range(n_inst):
            m = scenes._random_rigid(rng.uniform(0, 1, 3))
            m[:3, 3] = rng.uniform(-1.5, 1.5, 3) * scale
            inst[k, 0], inst[k, 1] = m.astype(np.float32), np.linalg.inv(m).astype(np.float32)
            tgt_sets.append(((objs[io[k]]["positions"].astype(np.float64) @ m[:3, :3].T + m[:3, 3]).astype(np.float32), objs[io[k]]["indices"]))
        inst[:, :, 3, :] = (0, 0, 0, 1)
        wv, wi, _s = make_mesh(rng)
        wv = (wv / np.float32(_s) * np.float32(scale)).astype(np.float32)
        sc = dict(objects=objs, instances=inst, instance_object=io, instance_material=np.full(n_inst, -1, dtype=np.int32),
                  world=dict(positions=wv, indices=wi, tri_material=np.zeros(len(wi), dtype=np.int32), tri_light=np.full(len(wi), -1, dtype=np.int32)),
                  materials=sc["materials"], lights=scenes._lights([]))
        tgt_sets.append((wv, wi))
        pick = rng.choice(len(tgt_sets), min(len(tgt_sets), 4), replace=False)
        rays = np.concatenate([aimed_rays(rng, tgt_sets[k][0], tgt_sets[k][1], scale, 60) for k in pick])
        idx = np.concatenate([o["indices"] for o in objs] + [wi])
    elif r_kind < 0.45:
        # TransformedPrimitive instances of the mesh (primitive.rs:105-159), overlapping; rays aimed at instanced vertices
        n_inst = int(rng.integers(1, 25))
        inst = np.zeros((n_inst, 2, 4, 4), dtype=np.float32)
        wv = []
        for k in range(n_inst):
            m = scenes._random_rigid(rng.uniform(0, 1, 3))
            m[:3, 3] = rng.uniform(-1.5, 1.5, 3) * scale
            inst[k, 0], inst[k, 1] = m.astype(np.float32), np.linalg.inv(m).astype(np.float32)
            wv.append((verts.astype(np.float64) @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
        inst[:, :, 3, :] = (0, 0, 0, 1)
        sc.update(instances=inst, instance_material=np.zeros(n_inst, dtype=np.int32))
        pick = rng.choice(n_inst, min(n_inst, 4), replace=False)
        rays = np.concatenate([aimed_rays(rng, wv[k], idx, scale, 60) for k in pick])
    else:
        rays = aimed_rays(rng, verts, idx, scale, 50_000 if BIG else 200)
        if rng.random() < 0.2:
            kw = dict(device_build=True)     # the tree built on the device (HLBVH, bvh.rs:475-811) against the oracle's HLBVH
    max_prims, split = int(rng.choice([1, 2, 4])), int(rng.choice([0, 0, 1, 2, 3]))
    if kw:
        split = pbrt_hip.SPLIT_HLBVH
    desc = (f"seed {seed}: {len(idx)} triangles, {n_inst} instances{' of ' + str(len(sc['objects'])) + ' objects + world triangles' if 'objects' in sc else ''}"
            f"{' (device-built tree)' if kw else ''}, scale {scale}, max_prims {max_prims}, split {split}, {len(rays)} rays")
    return sc, rays, max_prims, split, kw, n_inst, desc


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = pbrt_hip.Context(0)
    bad, n_wide, t0 = 0, 0, time.time()
    for seed in range(first, first + n_cases):
        sc, rays, max_prims, split, kw, n_inst, desc = make_case(seed)
        # round 5: how the wide records lie in HBM (pbrt_hip_context_set_wide_layout: by size / packed / one 64-byte line each);
        # its own generator, the draws of make_case stay what they were. The host-built copy below keeps the default layout:
        # the exported bytes are the packed form either way.
        layout = int(np.random.default_rng(seed ^ 0x11e5).integers(0, 3))
        desc += f" layout {layout}"
        try:
            osc = oracle.OracleScene(sc, max_prims, split)
            ctx.set_wide_layout(layout)
            try:
                gsc = pbrt_hip.Scene(ctx, sc, max_prims_in_node=max_prims, split_method=split, **kw)
            finally:
                ctx.set_wide_layout(pbrt_hip.WIDE_LAYOUT_AUTO)
            n_rec, why = gsc.wide_records()
            n_wide += n_rec > 0
            cpu, _ = osc.intersect(rays)
            occ = osc.intersect_p(rays)[0]
            got = {"default": (gsc.intersect(rays), gsc.intersect_p(rays))}
            ctx.set_counting(1)
            got["binary"] = (gsc.intersect(rays), gsc.intersect_p(rays))
            ctx.set_counting(0)
            if "instances" not in sc and "objects" not in sc and "spheres" not in sc:
                # the binary records walked without a stack (trace_stackless.h), on host- and device-built trees alike
                ctx.set_traversal(pbrt_hip.TRAVERSAL_STACKLESS)
                try:
                    got["stackless"] = (gsc.intersect(rays), gsc.intersect_p(rays))
                finally:
                    ctx.set_traversal(pbrt_hip.TRAVERSAL_AUTO)
            if n_rec >= 0 and "instances" not in sc and "objects" not in sc and seed % 3 == 0:
                # the device builder of the wide records (wide_gpu.hip) against the host builder (host_wide.cpp): same bytes
                ctx.set_wide_build(pbrt_hip.WIDE_BUILD_HOST)
                try:
                    hsc = pbrt_hip.Scene(ctx, sc, max_prims_in_node=max_prims, split_method=split, **kw)
                finally:
                    ctx.set_wide_build(pbrt_hip.WIDE_BUILD_DEVICE)
                n_t = len(sc["indices"])
                same = hsc.wide_records() == (n_rec, why) and all(a.tobytes() == b.tobytes() for a, b in zip(gsc.debug_wide_export(n_t), hsc.debug_wide_export(n_t)))
                hsc.close()
                if not same:
                    bad += 1
                    print(f"BUILDERS DIFFER {desc}: device {(n_rec, why)} host {hsc.wide_records() if False else '?'}", flush=True)
            gsc.close(); osc.close()
            for name, (h, p) in got.items():
                m = np.zeros(len(rays), dtype=bool)
                for f in ("prim_id", "t", "b0", "b1", "b2") + (("instance_id",) if n_inst and "instance_id" in h.dtype.names and "instance_id" in cpu.dtype.names else ()):
                    m |= (h[f] != cpu[f]) & ~(np.isnan(h[f].astype(np.float64)) & np.isnan(cpu[f].astype(np.float64)))
                m |= p != occ
                if m.any():
                    bad += 1
                    i = int(np.flatnonzero(m)[0])
                    print(f"MISMATCH ({name} kernel, wide records {n_rec} {why!r}) {desc}: {int(m.sum())} rays, first {i}: o {rays['o'][i]} d {rays['d'][i]} "
                          f"tmax {rays['t_max'][i]}\n   oracle {cpu[i]} occluded {occ[i]}\n   gpu    {h[i]} occluded {p[i]}", flush=True)
                    break
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"ERROR {desc}\n   {type(e).__name__}: {e}", flush=True)
        if (seed - first + 1) % 200 == 0:
            print(f"... {seed - first + 1} cases ({n_wide} with wide records), {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_intersect: {n_cases} cases from seed {first} ({n_wide} with wide records): {bad} mismatching", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
