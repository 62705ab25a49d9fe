"""Probe: do two concurrent half-frame renders (two contexts / streams, their k_shade and k_trace phases interleaving
on the GPU) beat one full-frame render? Wall time of both."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
if os.environ.get("PBRT_LIB"): pbrt_hip.LIB_PATH = os.environ["PBRT_LIB"]
from pbrt_hip import scenes
W, H = 1920, 1080
sc = scenes.random_triangles(1_000_000, seq=1)
cam = scenes.random_triangles_camera(W, H)
bvh = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, 0)
ctxs = [pbrt_hip.Context(0) for _ in range(2)]
gs = [pbrt_hip.Scene(c, sc, bvh=bvh) for c in ctxs]
import ctypes, numpy as np
films = [np.zeros((H, W, 4), dtype=np.float32) for _ in range(2)]


def render(i, spp, seed):
    return gs[i].render(cam, W, H, spp, max_depth=5, seed=seed)[1]


for it in range(2):
    t = time.perf_counter(); st = render(0, 64, 0); one = time.perf_counter() - t
rays1 = st["rays_closest"] + st["rays_shadow"]
for it in range(2):
    out = [None, None]
    def work(i):
        out[i] = render(i, 32, i)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    two = time.perf_counter() - t
rays2 = sum(o["rays_closest"] + o["rays_shadow"] for o in out)
print(f"one 64-spp render: {one*1e3:.1f} ms wall ({rays1/one/1e6:.0f} Mrays/s incl. host film copy); "
      f"two concurrent 32-spp renders: {two*1e3:.1f} ms wall ({rays2/two/1e6:.0f} Mrays/s)")
