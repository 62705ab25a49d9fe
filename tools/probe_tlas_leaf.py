import os, sys
sys.path.insert(0, "/root/repo/pbrt-rs_amd")
import numpy as np, pbrt_hip
from pbrt_hip import scenes
W, H, spp = 1920, 1080, 8
sc = scenes.instanced_scene(10_000, 1000)
cam = scenes.instanced_camera(W, H)
ctx = pbrt_hip.Context(0)
for tl in (4, 2, 1):
    blas_nodes, blas_order = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH)
    inst = pbrt_hip.make_instances(sc)
    lo, hi = pbrt_hip.instance_bounds(blas_nodes[0]["bmin"], blas_nodes[0]["bmax"], inst)
    tlas_nodes, tlas_order = pbrt_hip.bvh_build_boxes(lo, hi, tl, pbrt_hip.SPLIT_SAH)
    g = pbrt_hip.Scene(ctx, sc, bvh=(blas_nodes, blas_order, inst, tlas_nodes, tlas_order))
    for it in range(2):
        film, st = g.render(cam, W, H, spp, max_depth=16, seed=0)
    rays = st["rays_closest"] + st["rays_shadow"]
    ctx.set_counting(True); ctx.counters(reset=True)
    g.render(cam, W, H, spp, max_depth=16, seed=0)
    c = ctx.counters(reset=True); ctx.set_counting(False)
    print(f"tlas max_prims {tl}: {rays / st['total_ms'] / 1e3:.0f} Mrays/s; per ray node {c['node_tests']/c['rays']:.1f} "
          f"inst {c.get('inst_tests', 0)/c['rays']:.2f} tri {c['prim_tests']/c['rays']:.2f}", flush=True)
    g.close()
