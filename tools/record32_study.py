"""Host costing of a 32-byte, two-load wide record against the shipped 48-byte one (VERDICT r4 "Next round" 3): builds
config 3's tree with the product's host builder, draws a frame's ray mix with the CPU oracle standing in for the renderer
(camera rays; at every hit a light-sample shadow ray towards a uniform direction of the environment when it is above the
surface, a cosine-distributed MIS ray traced as a boolean query, a cosine-distributed continuation ray; depth 5), and runs
tools/native/record32_study.cpp over them. Host only.   usage: python tools/record32_study.py [n_triangles] [n_camera_rays]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle, pbrt_hip
from pbrt_hip import scenes

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_cam = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
W, H, DEPTH = 1920, 1080, 5
sc = scenes.random_triangles(n_tris, seq=1)
nodes, order = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH)
pos, idx = sc["positions"], sc["indices"][order]
tris = np.zeros((len(idx), 12), dtype=np.float32)          # the 48-B leaf-order records the builder reads: 9 vertex floats, prim, material, flags
tris[:, 0:3], tris[:, 3:6], tris[:, 6:9] = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
osc = oracle.OracleScene(sc)
rng = np.random.default_rng(5)

def rays_of(o, d, tmax=np.inf):
    r = np.zeros(len(o), dtype=scenes.RAY_DTYPE)
    r["o"], r["d"], r["t_max"] = o, d, tmax
    return r

def cosine_about(n, rng):
    u1, u2 = rng.uniform(0, 1, len(n)), rng.uniform(0, 1, len(n))
    r, ph = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    s = np.cross(n, a); s /= np.linalg.norm(s, axis=1, keepdims=True)
    t = np.cross(n, s)
    return s * (r * np.cos(ph))[:, None] + t * (r * np.sin(ph))[:, None] + n * np.sqrt(1 - u1)[:, None]

cam = scenes.camera_dict_to_floats(scenes.random_triangles_camera(W, H))
c2w, r2c = np.array(cam[:16]).reshape(4, 4), np.array(cam[16:32]).reshape(4, 4)
px = np.stack([rng.uniform(0, W, n_cam), rng.uniform(0, H, n_cam), np.zeros(n_cam), np.ones(n_cam)], 1)
pc = px @ r2c.T
pc = pc[:, :3] / pc[:, 3:4]
dc = pc / np.linalg.norm(pc, axis=1, keepdims=True)
cur = rays_of(np.tile(c2w[:3, 3], (n_cam, 1)), dc @ c2w[:3, :3].T)
out = []   # (rays, class)
cls_closest = 0
for bounce in range(DEPTH + 1):
    out.append((cur, cls_closest))
    cls_closest = 1
    hits, _ = osc.intersect(cur)
    ok = hits["prim_id"] >= 0
    if bounce == DEPTH or not ok.any():
        break
    o, d, t, prim = cur["o"][ok].astype(np.float64), cur["d"][ok].astype(np.float64), hits["t"][ok].astype(np.float64), hits["prim_id"][ok]
    p = o + d * t[:, None]
    tri = sc["indices"][prim]
    n = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]]).astype(np.float64)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n *= np.where((n * d).sum(1) > 0, -1.0, 1.0)[:, None]      # the side the ray arrived on
    p = p + n * 1e-5
    # uniform_sample_one_light on the one (environment) light: a uniform direction; the shadow ray is traced when f |cos| > 0
    z = rng.uniform(-1, 1, len(p)); ph = rng.uniform(0, 2 * np.pi, len(p)); rr = np.sqrt(1 - z * z)
    wi = np.stack([rr * np.cos(ph), rr * np.sin(ph), z], 1)
    up = (wi * n).sum(1) > 0
    out.append((rays_of(p[up], wi[up]), 2))
    out.append((rays_of(p, cosine_about(n, rng)), 3))            # estimate_direct's BSDF-sampled ray: a boolean query under an environment light
    keep = rng.uniform(0, 1, len(p)) < (1.0 if bounce < 3 else 0.5)   # matte rho 0.5: Russian roulette from the fourth bounce on
    cur = rays_of(p[keep], cosine_about(n[keep], rng))
flat = np.concatenate([np.concatenate([r["o"], r["d"], r["t_max"][:, None], np.full((len(r), 1), c, dtype=np.float32)], 1).astype(np.float32) for r, c in out])
counts = {c: sum(len(r) for r, cc in out if cc == c) for c in range(4)}
print(f"# rays: {len(flat)} = camera {counts[0]}, bounce closest-hit {counts[1]}, light-sample any-hit {counts[2]}, MIS boolean {counts[3]} "
      f"({len(flat) / n_cam:.2f} rays per camera sample; the measured frame has 4.86)", flush=True)
osc.close()
tmp = tempfile.mkdtemp(prefix="record32_")
exe = os.path.join(tmp, "record32_study")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "native", "record32_study.cpp"), "-o", exe, "-lpthread"])
np.ascontiguousarray(nodes).tofile(os.path.join(tmp, "nodes.bin")); tris.tofile(os.path.join(tmp, "tris.bin")); flat.tofile(os.path.join(tmp, "rays.bin"))
subprocess.check_call([exe, os.path.join(tmp, "nodes.bin"), os.path.join(tmp, "tris.bin"), os.path.join(tmp, "rays.bin")])
