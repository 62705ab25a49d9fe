"""One rank's share of the weak-scaling workload, timed on a single GPU: tile set `rank` of `world` at 64*world spp
(what bench.py --gpus world gives each rank), against the N=1 frame. Usage: probe_rank_of_world.py [world ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
from pbrt_hip import scenes

worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
strong = bool(os.environ.get("STRONG"))  # one frame of SPP (default 64) samples split over the ranks instead of 64 spp per rank
spp_strong = int(os.environ.get("SPP", "64"))  # 256 = BASELINE config 4, what bench.py --gpus N (N > 1) runs
W, H = 1920, 1080
ctx = pbrt_hip.Context(0)
sc = pbrt_hip.Scene(ctx, scenes.random_triangles(1_000_000, seq=1))
cam = scenes.random_triangles_camera(W, H)
for world in worlds:
    for rank in sorted({0, world - 1}):
        best = None
        for it in range(3):
            t0 = time.perf_counter()
            _, st = sc.render(cam, W, H, spp_strong if strong else 64 * world, max_depth=5, rr_threshold=1.0, light_strategy=1, seed=0,
                              tile_rank=rank, tile_world=world)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rays = st["rays_closest"] + st["rays_shadow"]
        print(f"world {world} rank {rank}: {rays / best / 1e6:8.1f} Mrays/s  {best * 1e3:7.1f} ms  "
              f"trace launches {st['trace_launches']}", flush=True)
