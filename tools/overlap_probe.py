"""Does a frame gain from two half-frames in flight at once? Two contexts (two streams) on one GPU, each renders the tiles of
one of two Morton shares of config 3 from its own host thread; wall time of both against one context rendering the whole frame.
While one stream's k_trace_wide runs (texture addressers busy, HBM at 15 %), the other's k_shade (HBM-bound) could fill in — if
the trace kernel's persistent grid leaves it registers: usage  LIB=variant.so python tools/overlap_probe.py. Dev tool."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
from pbrt_hip import scenes
if os.environ.get("LIB"):
    pbrt_hip.LIB_PATH = os.environ["LIB"]
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "64"))
N = int(os.environ.get("STREAMS", "2"))
sc = scenes.random_triangles(int(os.environ.get("TRIS", "1000000")), seq=1)
cam = scenes.random_triangles_camera(W, H)
import numpy as np
bvh = pbrt_hip.bvh_build(np.ascontiguousarray(sc['positions'], dtype=np.float32), np.ascontiguousarray(sc['indices'], dtype=np.int32))
ctxs = [pbrt_hip.Context(0) for _ in range(N)]
scs = [pbrt_hip.Scene(c, sc, bvh=bvh) for c in ctxs]


def whole():
    t = time.perf_counter()
    _, st = scs[0].render(cam, W, H, spp, max_depth=5, seed=0)
    return time.perf_counter() - t, st


def halves():
    out = [None] * N
    gate = threading.Barrier(N + 1)

    def work(r):
        gate.wait()
        out[r] = scs[r].render(cam, W, H, spp, max_depth=5, seed=0, tile_rank=r, tile_world=N)[1]
    th = [threading.Thread(target=work, args=(r,)) for r in range(N)]
    for t in th: t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    return time.perf_counter() - t0, out


whole(); halves()
for rep in range(3):
    tw, st = whole()
    th, sts = halves()
    rays = st["rays_closest"] + st["rays_shadow"]
    rays_h = sum(s["rays_closest"] + s["rays_shadow"] for s in sts)
    print(f"{os.path.basename(pbrt_hip.LIB_PATH)}: one stream {tw*1e3:.1f} ms wall ({st['total_ms']:.1f} device, trace {st['trace_ms']:.1f}) "
          f"{rays/tw/1e6:.0f} Mrays/s | {N} streams {th*1e3:.1f} ms wall {rays_h/th/1e6:.0f} Mrays/s "
          f"(device {' / '.join('%.1f' % s['total_ms'] for s in sts)}, trace {' / '.join('%.1f' % s['trace_ms'] for s in sts)})")
