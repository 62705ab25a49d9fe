"""Quick kernel-level timing of config 3 (one pass of 8 spp): trace / total ms. Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import pbrt_hip
from pbrt_hip import scenes
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    pbrt_hip.LIB_PATH = sys.argv[1]
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "8"))
sc = scenes.random_triangles(int(os.environ.get("TRIS", "1000000")), seq=1)
cam = scenes.random_triangles_camera(W, H)
ctx = pbrt_hip.Context(0)
layout = int(os.environ.get("LAYOUT", "0"))       # pbrt_hip.WIDE_LAYOUT_*: 0 auto, 1 packed (48-B stride), 2 one 64-byte line per record / triangle
ctx.set_wide_layout(layout)
scene = pbrt_hip.Scene(ctx, sc)
print("wide stride:", scene.wide_stride())
print("wide records:", scene.wide_records())
mode = int(os.environ.get("TRAVERSAL", "0"))     # pbrt_hip.TRAVERSAL_*: 0 wide records, 1 binary + stack, 2 binary stackless
ctx.set_traversal(mode)
print("traversal:", {0: "auto (4-wide records)", 1: "binary records, stack", 2: "binary records, stackless"}[mode])
for it in range(3):
    film, st = scene.render(cam, W, H, spp, max_depth=5, seed=0, samples_per_wave=int(os.environ.get("SPW", "0")))   # SPW: PbrtRenderParams.samples_per_wave
rays = st["rays_closest"] + st["rays_shadow"]
print(f"{os.path.basename(pbrt_hip.LIB_PATH)}: total {st['total_ms']:.1f} ms trace {st['trace_ms']:.1f} ms "
      f"({st['trace_launches']} launches) rays {rays/1e6:.1f}M -> {rays/st['total_ms']/1e3:.0f} Mrays/s, "
      f"trace-only {rays/st['trace_ms']/1e3:.0f} Mrays/s")
