"""Every rank's share of an N-GPU job, timed one after the other on a single GPU: what bounds the N-GPU rate before the film
reduce is the slowest rank (load_balance_max_over_mean of the bench line). Both dealing orders of the 16x16 tiles
(PbrtRenderParams.tile_order: Morton, SURVEY 8(e) — row-major, rounds 1-4) on BASELINE config 4 (1 M triangles, 1920x1080x256
spp) and config 5 (10 M instanced triangles, 3840x2160x128 spp in four 32-spp passes).
Usage: probe_rank_of_world.py [--configs 4 5] [--worlds 2 4 8] [--orders morton row-major] [--repeats 2]
(STRONG / SPP environment variables of rounds 1-2 are gone: the job is always the strong-scaling one bench.py --gpus N runs.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
sys.path.insert(0, ROOT)
import bench
import pbrt_hip
from pbrt_hip import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--configs", type=int, nargs="+", default=[4, 5])
ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--orders", nargs="+", default=["morton", "row-major"])
ap.add_argument("--repeats", type=int, default=2)
ap.add_argument("--spp-per-pass", type=int, default=-1, help="override the configuration's pass size (0 = sized from free HBM)")
a = ap.parse_args()
ctx = pbrt_hip.Context(0)
for cfg in a.configs:
    c = bench.CONFIGS[cfg]
    W, H = c["width"], c["height"]
    if a.spp_per_pass >= 0:
        c = dict(c, spp_per_pass=a.spp_per_pass)
    if c["scene"] == "instanced":
        sc = scenes.instanced_scene(c["tris"], c["instances"])
        scene = pbrt_hip.Scene(ctx, sc, bvh=pbrt_hip.build_two_level(sc))
        cam = scenes.instanced_camera(W, H)
    else:
        scene = pbrt_hip.Scene(ctx, scenes.random_triangles(c["tris"], seq=1))
        cam = scenes.random_triangles_camera(W, H)
    print(f"# config {cfg}: {W}x{H}x{c['spp']}spp, max_depth {c['max_depth']}, spp_per_pass {c['spp_per_pass'] or 'auto'}; best of {a.repeats} "
          f"renders per rank, HIP-event time of the render call (no film reduce)", flush=True)
    one = None
    for world in a.worlds:
        for order in (a.orders if world > 1 else a.orders[:1]):
            ms, rays = [], []
            for rank in range(world):
                best = None
                for _ in range(a.repeats):
                    _, st = scene.render(cam, W, H, c["spp"], max_depth=c["max_depth"], rr_threshold=1.0, light_strategy=1, seed=0,
                                         tile_rank=rank, tile_world=world, tile_order=0 if order == "morton" else 1, spp_per_pass=c["spp_per_pass"])
                    best = st["total_ms"] if best is None else min(best, st["total_ms"])
                ms.append(best)
                rays.append(st["rays_closest"] + st["rays_shadow"])
            mean = sum(ms) / len(ms)
            if world == 1:
                one = ms[0]
            print(f"config {cfg} world {world} {order:9s}: render ms max {max(ms):8.1f} mean {mean:8.1f} min {min(ms):8.1f}  "
                  f"load_balance_max_over_mean {max(ms) / mean:.4f}  rays max/mean {max(rays) / (sum(rays) / len(rays)):.4f}  "
                  f"slowest rank {sum(rays) / max(ms) / 1e3:8.1f} Mrays/s job rate before the reduce"
                  + (f" = {one / max(ms):.2f}x of one GPU ({one / max(ms) / world * 100:.0f} % of ideal)" if one and world > 1 else ""), flush=True)
    scene.close()
