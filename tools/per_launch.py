"""Per-launch durations of one config-3 frame from a rocprofv3 --kernel-trace CSV, next to PBRT_HIP_TRACE_LOG's ray
counts. Run as:  rocprofv3 --kernel-trace -d DIR -o per_launch --output-format csv -- python3 tools/per_launch.py render
then:            python3 tools/per_launch.py report DIR"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "render":
    sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
    import pbrt_hip
    from pbrt_hip import scenes
    W, H = 1920, 1080
    os.environ["PBRT_HIP_TRACE_LOG"] = "1"   # read once, by pbrt_hip_context_create: both frames are logged, the second follows "FRAME"
    ctx = pbrt_hip.Context(0)
    g = pbrt_hip.Scene(ctx, scenes.random_triangles(1_000_000, seq=1))
    cam = scenes.random_triangles_camera(W, H)
    for it in range(2):
        if it == 1:
            print("FRAME", file=sys.stderr, flush=True)
        _, st = g.render(cam, W, H, 64, max_depth=5, rr_threshold=1.0, light_strategy=1, seed=0,
                         tile_rank=0, tile_world=int(os.environ.get("TILE_WORLD", "1")))
    print(st, flush=True)
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:] if len(sys.argv) < 4 else rows
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"]
        name = name.split("(")[0].replace("void ", "")[:60]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > int(os.environ.get("MIN_NS", "50000")):
            print(f"{(s - t0) / 1e6:9.2f} ms  +{(e - s) / 1e6:8.3f} ms  {name}")
