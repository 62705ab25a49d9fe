"""Trace-only rate on a cache-resident scene vs config 3 (dev tool): how much of k_trace_wide's time is memory.
SCENE=cornell|rand20k|rand1m, SPP."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
from pbrt_hip import scenes
name = os.environ.get("SCENE", "cornell")
spp = int(os.environ.get("SPP", "16"))
if name == "cornell":
    sc, cam, W, H, depth = scenes.cornell_box(), scenes.cornell_camera(1920, 1080), 1920, 1080, 8
elif name == "rand20k":
    sc, cam, W, H, depth = scenes.random_triangles(20000, seq=3, size=0.05), scenes.random_triangles_camera(1920, 1080), 1920, 1080, 5
else:
    sc, cam, W, H, depth = scenes.random_triangles(1000000, seq=1), scenes.random_triangles_camera(1920, 1080), 1920, 1080, 5
ctx = pbrt_hip.Context(0)
g = pbrt_hip.Scene(ctx, sc)
for it in range(3):
    film, st = g.render(cam, W, H, spp, max_depth=depth, seed=0)
rays = st["rays_closest"] + st["rays_shadow"]
ctx.set_counting(2); ctx.wide_counters(reset=True); g.render(cam, W, H, spp, max_depth=depth, seed=0); wc = ctx.wide_counters(reset=True); ctx.set_counting(0)
print(f"{name} wide={g.wide_records()[0]}: trace {st['trace_ms']:.1f} ms ({st['trace_launches']} launches) rays {rays/1e6:.1f}M -> trace-only {rays/st['trace_ms']/1e3:.0f} Mrays/s, "
      f"frame {rays/st['total_ms']/1e3:.0f} Mrays/s; per ray {wc['records']/max(rays,1):.1f} records, {wc['triangles']/max(rays,1):.2f} triangles "
      f"-> {(wc['records']+wc['triangles'])/st['trace_ms']/1e6:.1f} G records/s")
