#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X path-tracing hot path on BASELINE.json's configs 3 and 4:
1 000 000 random triangles + constant env light, PathIntegrator max_depth 5, 1920x1080.

  python bench.py --gpus 1 --steps K --warmup W          config 3: 64 spp on one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
                                                          config 4: 256 spp in total, the 16x16 tiles of the ONE
                                                          frame dealt round-robin to the N ranks (total work fixed:
                                                          "scaling": "strong"), one RCCL reduce of the W*H*4 film to
                                                          rank 0 inside the timed step. --scaling weak renders
                                                          spp x N instead (per-GPU work fixed).

A step = one frame through Integrator::render (scene + BVH resident in HBM, film left on the device).
One ray = one Scene::intersect / intersect_p call. Prints ONE JSON line on rank 0. A rank that fails or hangs ends
the job with a non-zero exit code (the line, when there is one, is printed first).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
DEFAULTS = dict(width=1920, height=1080, tris=1_000_000, max_depth=5)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")
# what a stamped counter profile was measured on: the kernels, and what decides the workload they were given
KERNEL_SOURCES = ("pbrt-rs_amd/csrc", "pbrt-rs_amd/build.sh", "bench.py", "pbrt-rs_amd/pbrt_hip/scenes.py")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=DEFAULTS["width"])
    ap.add_argument("--height", type=int, default=DEFAULTS["height"])
    ap.add_argument("--spp", type=int, default=0,
                    help="samples per pixel of the frame (strong) or per GPU (weak); default 64 at one GPU, 256 in total at N > 1")
    ap.add_argument("--tris", type=int, default=DEFAULTS["tris"])
    ap.add_argument("--max-depth", type=int, default=DEFAULTS["max_depth"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[640, 360], help="crop rendered by the CPU oracle")
    ap.add_argument("--cpu-spp", type=int, default=48)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default): the tiles of ONE frame are split over the GPUs (BASELINE config 4); "
                         "weak: the frame has spp x N samples per pixel (per-GPU work fixed)")
    ap.add_argument("--abi-reduce-check", action="store_true",
                    help="after the measurement, repeat the film merge untimed through the C ABI's own RCCL communicator "
                         "(pbrt_hip_comm_create / pbrt_hip_film_reduce) and compare with torch.distributed's; needs "
                         "torch.distributed.run. Not part of the default job: the measured run does not depend on a second communicator")
    ap.add_argument("--watchdog-s", type=float, default=600.0, help="a rank stuck longer than this ends the job with exit code 3")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl (= RCCL over xGMI, the measured job) or gloo: the same job with the film reduce and the agreement "
                         "collectives staged through host memory — the rehearsal of the N > 1 path on a box with ONE GPU, where "
                         "RCCL refuses two ranks on one device (tests/test_gpu_two_ranks.py)")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses device 0 (rehearsal on a one-GPU box, with --dist-backend gloo)")
    ap.add_argument("--save-film", default="", help="rank 0 writes the reduced film of the last step here (.npy)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed config-5 block (N = 1 only)")
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except Exception:
            pass
    return n


def algorithmic_bytes(rays_closest, rays_shadow, node_tests, prim_tests):
    """32 B ray in + 32 B per box test + 48 B per triangle test + 16 B (closest) or 4 B (any) out."""
    return 32 * (rays_closest + rays_shadow) + 32 * node_tests + 48 * prim_tests + 16 * rays_closest + 4 * rays_shadow


def kernel_source_hash():
    """sha1 over the kernel sources (pbrt-rs_amd/csrc/*, build.sh), this file and the scene generator: what the stamped counter
    profile was measured on."""
    import hashlib
    h = hashlib.sha1()
    files = []
    for rel in KERNEL_SOURCES:
        path = os.path.join(ROOT, rel)
        if os.path.isdir(path):
            files += [os.path.join(path, f) for f in os.listdir(path) if f.endswith((".h", ".hip", ".cpp"))]
        elif os.path.exists(path):
            files.append(path)
    for f in sorted(files):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def measured_traffic(args, world, spp_total, kernel):
    """The stamped counter profile of this very command (profiles/r04_traffic.json, written by tools/measure_traffic.sh:
    rocprofv3 --kernel-trace --stats and separate --pmc passes). It was measured for ONE configuration and ONE state of
    the kernel sources: returned only when this run is that configuration (otherwise None and the reason); `stale` says
    that the kernel sources have changed since (the figures are then reported as measured, flagged)."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except Exception:
        return None, "profiles/r04_traffic.json missing", False
    mine = dict(n_gpus=world, tris=args.tris, width=args.width, height=args.height, spp=spp_total, max_depth=args.max_depth, kernel=kernel)
    diff = {k: (v, rec["config"].get(k)) for k, v in mine.items() if rec["config"].get(k) != v}
    if diff:
        return None, f"measured for another configuration: {diff}", False
    return rec, None, rec.get("source_hash") != kernel_source_hash()


def loaded_runtime_libs():
    """Which HIP runtime / RCCL this process ended up on (/proc/self/maps): a Python host holds PyTorch's bundled ROCm and
    /opt/rocm side by side; the first to initialise owns the GPU."""
    libs = set()
    try:
        for line in open("/proc/self/maps"):
            path = line.rsplit(" ", 1)[-1].strip()
            base = os.path.basename(path)
            if base.startswith(("libamdhip64", "librccl", "libhsa-runtime64", "libpbrt_hip")):
                libs.add(path)
    except OSError:
        pass
    return sorted(libs)


def rank_report(torch, device, local_rank, rank, stats, elapsed_s):
    """What one rank of the N-GPU job knows about itself (gathered into config.ranks): which device it really ran on, how
    long ITS renders took (HIP events around the whole render call, per step) and how many rays it traced — load balance
    is what bounds the N-GPU rate, and a line built on N ranks that shared one device would not be a scaling number."""
    props = torch.cuda.get_device_properties(device)
    free_b, total_b = torch.cuda.mem_get_info(device)
    ms = [st["total_ms"] for st in stats] or [0.0]
    return {
        "rank": rank, "local_rank": local_rank, "device": props.name,
        "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}",
        "hbm_free_GB": round(free_b / 1e9, 1), "hbm_total_GB": round(total_b / 1e9, 1),
        "render_ms_per_step": {"mean": round(sum(ms) / len(ms), 2), "max": round(max(ms), 2)},
        "trace_ms_per_step": round(sum(st["trace_ms"] for st in stats) / max(len(stats), 1), 2),
        "rays_per_step": int(sum(st["rays_closest"] + st["rays_shadow"] for st in stats) / max(len(stats), 1)),
        "wall_s": round(elapsed_s, 4), "runtime_libs": loaded_runtime_libs(), "pid": os.getpid(),
    }


def gather_rank_reports(dist, use_dist, world, report):
    """Collective (all_gather_object): every rank's report, in rank order, on every rank."""
    if not use_dist:
        return [report]
    reports = [None] * world
    dist.all_gather_object(reports, report)
    return reports


def check_distinct_devices(reports, one_gpu):
    """N ranks must be N devices (uuid, or PCI bus id where the runtime reports no uuid) unless the job was started as the
    one-GPU rehearsal (--one-gpu)."""
    ids = [r.get("uuid") or r.get("pci_bus_id") for r in reports]
    if len(set(ids)) != len(ids) and not one_gpu:
        raise RuntimeError(f"{len(ids)} ranks on {len(set(ids))} distinct device(s): {ids} (use --one-gpu for the one-GPU rehearsal)")
    return len(set(ids))


def load_balance(reports):
    """max / mean of the ranks' own render time per step: 1.0 = perfectly even; the N-GPU rate is bounded by 1 / this."""
    ms = [r["render_ms_per_step"]["mean"] for r in reports]
    mean = sum(ms) / len(ms)
    return round(max(ms) / mean, 4) if mean > 0 else None


class Stage:
    """Collective-safe error handling: every rank reports after each stage whether it got through; if any rank did
    not, all of them leave together (non-zero) instead of some waiting in the next collective for a rank that is gone."""

    def __init__(self, dist, torch, device, use_dist, rank):
        self.dist, self.torch, self.device, self.use_dist, self.rank = dist, torch, device, use_dist, rank

    def run(self, name, fn):
        """fn must not contain a collective another rank could miss (a rank failing before it would leave the others
        waiting inside): collectives go into stages of their own, or, in the measured loop, behind agree()."""
        err, out = None, None
        try:
            self.inject(name)
            out = fn()
        except StageFailed:
            raise
        except BaseException as e:  # noqa: BLE001 - reported and turned into an exit code
            err = f"{type(e).__name__}: {e}"
        self.agree(name, err)
        return out

    def collective(self, name, fn):
        """fn IS a collective (reduce, broadcast, communicator creation ...). A rank that fails in it cannot tell the
        others through another collective: it leaves at once, non-zero; the peers see the broken connection (or the
        launcher ends them, or their watchdog does)."""
        try:
            self.inject(name)
            return fn()
        except BaseException as e:  # noqa: BLE001
            print(f"[bench rank {self.rank}] collective '{name}' failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            raise StageFailed(name, f"{type(e).__name__}: {e}")

    def inject(self, name):
        if os.environ.get("PBRT_BENCH_FAIL") == f"{name}@{self.rank}":  # test hook: this rank fails at this point
            raise RuntimeError("injected failure (PBRT_BENCH_FAIL)")

    def agree(self, name, err):
        """Collective: every rank says whether it got through `name`; if one did not, all raise StageFailed."""
        ok = self.torch.tensor([0.0 if err else 1.0], device=self.device)
        if self.use_dist:
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if float(ok[0]) < 1.0:
            print(f"[bench rank {self.rank}] stage '{name}' failed on {'this rank: ' + err if err else 'another rank'}",
                  file=sys.stderr, flush=True)
            raise StageFailed(name, err)


class StageFailed(RuntimeError):
    pass


def run_steps(render_into, films, steps, warmup, dist, use_dist, sync, make_event=None, stage=None, reduce_film=None):
    """The measured loop, shared by the GPU job below and by the two-rank CPU test (tests/test_multi_gpu_gloo.py, gloo):
    W untimed and K timed steps, a step = render this rank's tile share of the frame into a film, then (N > 1) reduce the
    film to rank 0 — the only collective. Films alternate so that the reduce of frame k may overlap the render of frame
    k + 1; a film is reused only after the reduce that read it has finished (make_event() returns an object with
    synchronize(), recorded after the reduce; None where the reduce is synchronous). Bracketed by barrier + sync on both
    sides. In the untimed warm-up steps (and in the first one at least) the ranks agree after rendering that all of them
    got through before any enters the reduce: a rank whose render fails is then seen, not waited for. reduce_film(film)
    replaces the plain dist.reduce where the film cannot be handed to the backend as it is (gloo + device films). The timed steps
    carry no such exchange; a rank that fails there exits non-zero and the launcher ends the others (a rank that hangs:
    the watchdog). Returns (seconds, per-step stats list, the film of the last step)."""
    reduced = [None] * len(films)
    frame = [0]

    def step(checked=False):
        k = frame[0] % len(films)
        frame[0] += 1
        if reduced[k] is not None:
            reduced[k].synchronize()
        if checked and stage is not None:
            err, st = None, None
            try:
                stage.inject("render")
                st = render_into(films[k])
            except BaseException as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            stage.agree("render", err)
        else:
            st = render_into(films[k])
        if use_dist:
            if reduce_film is not None:
                reduce_film(films[k])
            else:
                dist.reduce(films[k], dst=0, op=dist.ReduceOp.SUM)
            reduced[k] = make_event() if make_event else None
        return st

    for _ in range(warmup):
        step(checked=True)
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    stats = [step() for _ in range(steps)]
    sync()
    if use_dist:
        dist.barrier()
    return time.perf_counter() - t0, stats, films[(frame[0] - 1) % len(films)]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch
    import torch.distributed as dist
    import pbrt_hip
    from pbrt_hip import scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible (the HIP path has no CPU fallback)")
    if args.one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the RCCL path runs even with one rank
    staged = args.dist_backend == "gloo"            # collectives on host tensors; films staged through host memory
    coll_device = torch.device("cpu") if staged else device

    out = {}
    printed = [False]

    def emit():
        if rank == 0 and out and not printed[0]:
            printed[0] = True
            print(json.dumps(out), flush=True)

    # A stuck communicator (or kernel) must not look like success: the watchdog prints what there is and ends this
    # rank with exit code 3; torch.distributed.run then tears the other ranks down.
    def give_up():
        print(f"[bench rank {rank}] watchdog: no progress for {args.watchdog_s:.0f} s, giving up", file=sys.stderr, flush=True)
        if out:
            out.setdefault("error", "watchdog timeout")
        emit()
        os._exit(3)

    dog = threading.Timer(args.watchdog_s, give_up)
    dog.daemon = True
    dog.start()

    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if staged:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    stage = Stage(dist, torch, coll_device, use_dist, rank)

    def barrier():
        if use_dist:
            dist.barrier()

    W, H = args.width, args.height
    if args.spp <= 0:
        args.spp = 64 if (world == 1 or args.scaling == "weak") else 256
    spp_total = args.spp * world if args.scaling == "weak" else args.spp
    config_name = "config3" if (world == 1 and spp_total == 64) else ("config4" if spp_total == 256 and args.scaling == "strong" else "config3/4 variant")
    rc = 0
    try:
        def setup():
            sc = scenes.random_triangles(args.tris, seq=1)
            cam = scenes.random_triangles_camera(W, H)
            t0 = time.time()
            bvh = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH)
            t_bvh = time.time() - t0
            ctx = pbrt_hip.Context(local_rank)
            scene = pbrt_hip.Scene(ctx, sc, bvh=bvh)
            return sc, cam, t_bvh, ctx, scene

        sc, cam, t_bvh, ctx, scene = stage.run("scene", setup)
        n_wide, wide_reason = scene.wide_records()
        kernel = "k_trace_wide" if n_wide >= 0 else "k_trace"
        # Two films: the reduce of frame k (torch's stream) may still be reading its film while frame k+1 is rendered
        # (the library's stream)
        films = [torch.zeros((H, W, 4), dtype=torch.float32, device=device) for _ in range(2 if use_dist else 1)]

        def render_into(film, **kw):
            return scene.render(cam, W, H, spp_total, max_depth=args.max_depth, rr_threshold=1.0, light_strategy=1, seed=0,
                                tile_rank=rank, tile_world=world, spp_per_pass=args.spp_per_pass, d_film_ptr=film.data_ptr(), **kw)[1]

        def record_event():
            e = torch.cuda.Event()
            e.record()
            return e

        def reduce_staged(film):
            # gloo: the film goes through host memory (the render call returned after draining the library's stream)
            host = film.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                film.copy_(host)

        def timed():
            seconds, stats, last = run_steps(render_into, films, args.steps, args.warmup, dist, use_dist, torch.cuda.synchronize, record_event, stage,
                                             reduce_staged if (staged and use_dist) else None)
            if args.save_film and rank == 0:
                np.save(args.save_film, last.cpu().numpy())
            return (seconds, sum(st["rays_closest"] + st["rays_shadow"] for st in stats), sum(st["trace_ms"] for st in stats),
                    sum(st["trace_launches"] for st in stats), stats)

        elapsed, rays, trace_ms, trace_launches, step_stats = timed()   # carries its own agreement points (run_steps)
        my_rays, my_elapsed = rays, elapsed
        # every rank's own account of the run (device identity, its render times, its rays): config.ranks
        report = stage.run("rank report", lambda: rank_report(torch, device, local_rank, rank, step_stats, elapsed))
        reports = stage.collective("gather rank reports", lambda: gather_rank_reports(dist, use_dist, world, report))
        n_devices = stage.run("distinct devices", lambda: check_distinct_devices(reports, args.one_gpu))
        if use_dist:
            def gather_times():
                tt = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=coll_device)
                tmax = tt.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                return float(tmax[0]), float(tt[1])
            elapsed, rays = stage.collective("max over ranks", gather_times)
        value = rays / elapsed / 1e6

        if rank == 0:
            launches_per_frame = trace_launches / args.steps
            trace_s_per_launch = trace_ms * 1e-3 / max(trace_launches, 1)
            # ---- untimed, instrumented renders of this rank's tile set ----
            # (1) the reference's loops: box / triangle tests of BVHAccel::intersect for exactly these rays -> SURVEY 8(d)
            ctx.set_counting(1)
            ctx.counters(reset=True)
            st_c = render_into(films[0])
            c = ctx.counters(reset=True)
            frame_rays = st_c["rays_closest"] + st_c["rays_shadow"]
            alg_bytes = algorithmic_bytes(st_c["rays_closest"], st_c["rays_shadow"], c["node_tests"], c["prim_tests"])
            # (2) what the kernel itself fetches: 48-B records and 48-B triangles, three 16-B lane requests each
            wc = None
            if n_wide >= 0:
                ctx.set_counting(2)
                ctx.wide_counters(reset=True)
                render_into(films[0])
                wc = ctx.wide_counters(reset=True)
            ctx.set_counting(0)
            if wc is not None:
                rec_per_launch = (wc["records"] + wc["triangles"]) / max(launches_per_frame, 1)
                rec_bytes, table_bytes, waves = 48, max(n_wide, 1) * 48, 5
            else:   # binary child-pair records: one 64-B record per two box tests
                rec_per_launch = (c["node_tests"] / 2 + c["prim_tests"]) / max(launches_per_frame, 1)
                rec_bytes, table_bytes, waves = 64, scene_interior_bytes(scene), 6
            achieved_rec = rec_per_launch / trace_s_per_launch / 1e9
            # ---- measured ceilings of the fetch pattern (dependent random record fetches, nothing else to do) ----
            ceil = {
                "same_footprint_kernel_occupancy": ctx.probe_gather(table_bytes, rec_bytes, waves) / 1e9,
                "same_footprint_8_waves": ctx.probe_gather(table_bytes, rec_bytes, 8) / 1e9,
                "l2_resident_8_waves": ctx.probe_gather(2 << 20, rec_bytes, 8) / 1e9,
                "l1_resident_8_waves": ctx.probe_gather(16 << 10, rec_bytes, 8) / 1e9,
            }
            peak_rec = ceil["l1_resident_8_waves"]
            # ---- the stamped counter profile of this command (HBM bytes, issue, texture addressers) ----
            tri_bytes = args.tris * 48
            compulsory = 32 * frame_rays + 16 * st_c["rays_closest"] + 4 * st_c["rays_shadow"] + 4 * frame_rays   # rays in, hits out, queue
            compulsory_per_launch = compulsory / max(launches_per_frame, 1) + (table_bytes + tri_bytes)          # + the tree, once
            rec, why_not, stale = measured_traffic(args, world, spp_total, kernel)
            tr = rec["trace"] if rec else None
            # the stamped counters belong to the launches they were collected on: a run whose own launches take a different time
            # (another box, clock, driver) is not described by them either
            stamped_launch_ms = tr["avg_launch_ns_under_kernel_trace"] * 1e-6 if tr and tr.get("avg_launch_ns_under_kernel_trace") else None
            drift = (trace_s_per_launch * 1e3 / stamped_launch_ms - 1.0) if stamped_launch_ms else None
            if rec and drift is not None and abs(drift) > 0.05:
                stale = True
            source = (f"{os.path.relpath(TRAFFIC_FILE, ROOT)}: rocprofv3 --pmc passes over this command, commit {rec['commit']}, "
                      f"sources {rec.get('source_hash')}" + (" (STALE: the sources have changed since, or this run's launches take "
                      f"{drift * 100:+.1f} % of the profiled ones')" if stale else "")) if rec else why_not
            traffic = tr["bytes_per_launch"] if tr else None
            hbm_frac = traffic / trace_s_per_launch / 1e9 / HBM_PEAK_GBS if traffic else None
            peak_ips = 256 * 4 * 2.4e9 / 2   # one wave64 vector instruction per two cycles per SIMD, 1024 SIMDs at 2.4 GHz
            valu_frac = tr["valu_insts_per_launch"] / trace_s_per_launch / peak_ips if tr and tr.get("valu_insts_per_launch") else None
            ta_frac = tr.get("ta_busy_fraction") if tr else None
            # Headline (frozen in round 3): the busiest hardware unit of the dominant kernel, from the
            # stamped counters — the largest of {texture addressers busy, vector issue rate, HBM bytes / peak}. Everything
            # else in this block is a diagnostic next to it.
            units = {"TA": ta_frac, "VALU issue": valu_frac, "HBM": hbm_frac}
            known = {k: v for k, v in units.items() if v is not None}
            bound = max(known, key=known.get) if known else None
            if bound == "TA":
                head = {"achieved": round(tr["ta_busy_cycles_per_launch"]), "peak": round(tr["gpu_cycles_per_launch"]),
                        "unit": "busy cycles per launch, mean of the 256 texture addressers, against the kernel's GPU cycles"}
            elif bound == "VALU issue":
                head = {"achieved": round(tr["valu_insts_per_launch"] / trace_s_per_launch / 1e9, 1), "peak": round(peak_ips / 1e9, 1),
                        "unit": "G wave-instructions/s"}
            elif bound == "HBM":
                head = {"achieved": round(traffic / trace_s_per_launch / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            else:
                head = {"achieved": None, "peak": None, "unit": None}
            roofline = {
                "kernel": kernel,
                "bound": bound, **head, "frac": round(known[bound], 4) if bound else None,
                "definition": "busiest hardware unit of the dominant kernel = max(TA busy, VALU issue, HBM bytes / 8 TB/s), stamped rocprofv3 counters",
                # bound / achieved / peak / frac / traffic / units come from the STAMPED profile (counters cannot be read inside a
                # run); what this run itself measured is launch time (HIP events) and the gather / algorithmic / wide blocks below
                "measured_in_this_run": False, "stamped_avg_launch_ms": round(stamped_launch_ms, 4) if stamped_launch_ms else None,
                "launch_time_vs_stamped": round(drift, 4) if drift is not None else None,
                "counter_derived": None if rec else f"null: {why_not} (bound, achieved, peak, frac, traffic, units, hbm.frac); gather / algorithmic / wide are this run's own, rank 0's tile share",
                "traffic": traffic, "stale": stale if rec else None, "source": source,
                "units": {k: (round(v, 4) if v is not None else None) for k, v in units.items()},
                "valu_lane_utilisation": round(tr["valu_lane_utilisation"], 3) if tr and tr.get("valu_lane_utilisation") else None,
                "waves_waiting_fraction": round(tr["wave_cycles_waiting_fraction"], 3) if tr and tr.get("wave_cycles_waiting_fraction") else None,
                "hbm": {
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "compulsory_bytes_per_launch": round(compulsory_per_launch),
                    "compulsory_GBps": round(compulsory_per_launch / trace_s_per_launch / 1e9, 1),
                    "counter_bytes_per_launch": traffic,
                    "achieved": round(traffic / trace_s_per_launch / 1e9, 1) if traffic else None,
                    "frac": round(hbm_frac, 4) if hbm_frac is not None else None,
                },
                # diagnostic: the kernel's own fetches (48-B records + triangles, counted by k_trace_wide<COUNT>) against dependent
                # record fetches from an L1-resident table, both measured in this run (pbrt_hip_probe_gather)
                "gather": {"achieved": round(achieved_rec, 2), "peak": round(peak_rec, 2), "unit": "G records/s",
                           "frac": round(achieved_rec / peak_rec, 4), "record_bytes": rec_bytes, "records_per_launch": round(rec_per_launch),
                           "table_bytes": table_bytes, "ceilings_G_records_per_s": {k: round(v, 2) for k, v in ceil.items()}},
                # SURVEY 8(d)'s figure, kept as a reported quantity: the bytes the REFERENCE's loop touches for these rays
                "algorithmic": {
                    "bytes_per_launch": round(alg_bytes / max(launches_per_frame, 1)),
                    "GBps": round(alg_bytes / max(launches_per_frame, 1) / trace_s_per_launch / 1e9, 1),
                    "bytes_per_ray": round(alg_bytes / max(frame_rays, 1), 1),
                    "node_tests_per_ray": round(c["node_tests"] / max(c["rays"], 1), 2),
                    "tri_tests_per_ray": round(c["prim_tests"] / max(c["rays"], 1), 2),
                    "note": "served mostly from L2 / Infinity Cache: not a fraction of HBM peak",
                },
                "launches_per_step": launches_per_frame, "avg_launch_ms": round(trace_s_per_launch * 1e3, 4),
                "trace_fraction_of_step": round(trace_ms * 1e-3 / args.steps / (my_elapsed / args.steps), 3),
            }
            sh = rec.get("shade") if rec else None
            if sh and sh.get("avg_launch_ns_under_kernel_trace"):
                # the second kernel of the frame: k_shade streams the path state (stamped profile; its launch time is the
                # profile's own, the library times only the traversal launches)
                # FETCH_SIZE under-counts the coalesced SoA streams k_shade reads: calibrated on that pattern (factor and file below)
                shade_bytes = sh.get("bytes_per_launch_calibrated") or sh["bytes_per_launch"]
                gbps = shade_bytes / sh["avg_launch_ns_under_kernel_trace"]
                roofline["shade"] = {"kernel": "k_shade", "bound": "HBM", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gbps / HBM_PEAK_GBS, 4), "traffic": shade_bytes, "traffic_counters_at_face_value": sh["bytes_per_launch"],
                                     "source": (f"FETCH_SIZE / {sh['fetch_calibration']['factor']} + WRITE_SIZE: {sh['fetch_calibration']['source']}"
                                                if sh.get("fetch_calibration") else "counters at face value (uncalibrated)"),
                                     "measured_in_this_run": False,
                                     "avg_launch_ms": round(sh["avg_launch_ns_under_kernel_trace"] * 1e-6, 4), "launches_per_step": sh.get("launches_per_step"),
                                     "valu_lane_utilisation": sh.get("valu_lane_utilisation"), "stale": stale}
            if wc is not None:
                roofline["wide"] = {"records_per_ray": round(wc["records"] / max(frame_rays, 1), 2),
                                    "leaf_candidates_per_ray": round(wc["leaf_candidates"] / max(frame_rays, 1), 2),
                                    "triangles_per_ray": round(wc["triangles"] / max(frame_rays, 1), 2),
                                    "rays_left_to_binary_kernel": wc["special_rays"], "n_records": n_wide}
            secondary = None
            config4_n1 = None
            if world == 1 and not args.no_secondary:
                try:   # reported beside the measurement, never instead of it
                    secondary = secondary_config5(torch, pbrt_hip, scenes, ctx, device, peak_rec)
                except Exception as e:  # noqa: BLE001
                    secondary = {"error": f"{type(e).__name__}: {e}"}
                if config_name == "config3":
                    # The N > 1 job is BASELINE config 4 (256 spp, the tiles of the one frame split over the ranks): its
                    # one-GPU point, so that the 1 / 2 / 4 / 8 curve has an anchor on the same workload. Untimed for `value`.
                    try:
                        runs4 = [scene.render(cam, W, H, 256, max_depth=args.max_depth, rr_threshold=1.0, light_strategy=1, seed=0,
                                              d_film_ptr=films[0].data_ptr())[1] for _ in range(2)]
                        ms4 = [r["total_ms"] for r in runs4]
                        rays4 = runs4[0]["rays_closest"] + runs4[0]["rays_shadow"]
                        config4_n1 = {"workload": f"config4 on ONE GPU: the same scene, {W}x{H}x256spp (what --gpus N splits over N ranks); "
                                                  f"mean of {len(ms4)} frames, HIP-event time of the render call, no film reduce; not part of `value`",
                                      "value": round(rays4 / (sum(ms4) / len(ms4)) / 1e3, 1), "unit": "Mrays/s",
                                      "ms_per_frame": round(sum(ms4) / len(ms4), 2), "rays_per_frame": int(rays4)}
                    except Exception as e:  # noqa: BLE001
                        config4_n1 = {"error": f"{type(e).__name__}: {e}"}
            cpu_baseline = None
            if world == 1 and not args.no_cpu_baseline:
                # ---- CPU baseline: the oracle (C++ restatement; the Rust reference cannot be built) on a
                # bounded crop of the same frame, all host cores this process may use ----
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle
                cores = host_cores()
                cw, ch = args.cpu_crop
                x0, y0 = (W - cw) // 2, (H - ch) // 2
                osc = oracle.OracleScene(sc)
                _, st_o = osc.render(scenes.camera_dict_to_floats(cam), W, H, args.cpu_spp, max_depth=args.max_depth,
                                     rr_threshold=1.0, light_strategy=1, seed=0, bounds=(x0, y0, x0 + cw, y0 + ch),
                                     n_threads=cores)
                cpu_baseline = {
                    "value": round(st_o["rays"] / st_o["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": cores,
                    "kind": "port",
                    "sample": f"{cw}x{ch} centre crop of the {W}x{H} frame at {args.cpu_spp} spp, same scene/seed, "
                              f"{cores} oracle threads ({st_o['rays']} rays in {st_o['seconds']:.1f} s)",
                }
                osc.close()
            split = "one GPU" if world == 1 else (f"{args.spp} spp per GPU" if args.scaling == "weak" else "tiles of the one frame split over the GPUs")
            out.update({
                "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
                "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {
                    "workload": f"{config_name}: {args.tris} random triangles + constant env light, PathIntegrator "
                                f"max_depth {args.max_depth}, {W}x{H}x{spp_total}spp ({split}), "
                                f"SAH BVH <=4 prims/leaf, seed 0",
                    "parallelism": (f"tiles16x16 round-robin over {world} rank(s); "
                                    + ("film reduce staged through host memory (gloo), all ranks on GPU 0: a rehearsal, not a scaling number"
                                       if staged else "RCCL film reduce") + " inside the step") if world > 1 else "1 GPU",
                    "sec_per_frame": round(elapsed / args.steps, 4),
                    "rays_per_frame": int(rays / args.steps),
                    "bvh_build_s_host": round(t_bvh, 2),
                    "traversal": f"{kernel}" + (f" ({n_wide} 4-wide records)" if n_wide >= 0 else f" (binary records: {wide_reason})"),
                },
                "roofline": roofline, "cpu_baseline": cpu_baseline, "secondary": secondary, "config4_n1": config4_n1,
            })
            out["config"]["dist_backend"] = args.dist_backend if use_dist else None
            out["config"]["runtime_libs"] = loaded_runtime_libs()
            # per rank: device identity, its own render time per step, its rays (DESIGN.md section 6 says how to read them)
            out["config"]["ranks"] = reports
            out["config"]["n_devices"] = n_devices
            out["config"]["load_balance_max_over_mean"] = load_balance(reports)

        if args.abi_reduce_check and use_dist and not staged:
            # Untimed: the same film merge through the C ABI's own RCCL communicator (pbrt_hip_comm_create /
            # pbrt_hip_film_reduce), checked against torch.distributed's reduce. Every rank reports after each stage.
            f = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
            stage.run("abi: render", lambda: render_into(f))
            ref = f.clone()
            stage.collective("abi: torch reduce", lambda: dist.reduce(ref, dst=0, op=dist.ReduceOp.SUM))

            def share_id():
                ids = [pbrt_hip.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0, device=device)
                torch.cuda.synchronize()
                return ids[0]
            uid = stage.collective("abi: id broadcast", share_id)
            comm = stage.collective("abi: comm_create", lambda: pbrt_hip.Comm(ctx, world, rank, uid))
            stage.collective("abi: film_reduce", lambda: comm.film_reduce(f.data_ptr(), W * H, root=0))
            stage.run("abi: comm_destroy", comm.close)

            def compare():
                if rank != 0:
                    return "ok"
                diff, scale = float((f - ref).abs().max()), float(ref.abs().max())
                msg = f"max |abi - torch| = {diff:.3g} of {scale:.3g}"
                if not diff <= 1e-5 * scale:
                    raise RuntimeError("MISMATCH: " + msg)
                return "ok: " + msg
            status = stage.run("abi: compare", compare)
            if rank == 0:
                out["config"]["abi_film_reduce"] = status
        emit()
        barrier()
        scene.close()
        ctx.close()
    except StageFailed as e:
        if rank == 0 and out:
            out["error"] = f"stage '{e.args[0]}' failed" + (f": {e.args[1]}" if e.args[1] else " on another rank")
        emit()
        rc = 4
    finally:
        dog.cancel()
    if use_dist and rc == 0:
        dist.destroy_process_group()
    if rc:
        sys.stdout.flush()
        os._exit(rc)  # a failed job must not wait in destroy_process_group for ranks that are gone


def secondary_config5(torch, pbrt_hip, scenes, ctx, device, peak_rec):
    """Untimed for `value`: BASELINE config 5's scene (10 000 base triangles x 1000 rigid instances, matte / mirror / glass by
    instance, env light, PathIntegrator depth 16) at its stated 3840x2160, 32 of the 128 spp = ONE PASS of the full job (2^28
    concurrent paths: the 128-spp frame is four such passes; with 8 spp the wavefronts are a quarter as long and the rate reads
    5 % lower, profiles/r04_wide_kernel_ladder.txt), through pbrt_hip_render_device — the two-level traversal kernel k_trace_wide<false, 1> (primitive.rs:136-159)."""
    W5, H5, SPP5, DEPTH5 = 3840, 2160, 32, 16
    sc5 = scenes.instanced_scene(10_000, 1000)
    scene5 = pbrt_hip.Scene(ctx, sc5, bvh=pbrt_hip.build_two_level(sc5))
    cam5 = scenes.instanced_camera(W5, H5)
    film5 = torch.zeros((H5, W5, 4), dtype=torch.float32, device=device)

    def go():
        return scene5.render(cam5, W5, H5, SPP5, max_depth=DEPTH5, rr_threshold=1.0, light_strategy=1, seed=0, d_film_ptr=film5.data_ptr())[1]
    go()
    runs = [go() for _ in range(3)]
    best = min(runs, key=lambda r: r["total_ms"])
    st = dict(best, total_ms=sum(r["total_ms"] for r in runs) / len(runs), trace_ms=sum(r["trace_ms"] for r in runs) / len(runs))   # the MEAN frame
    rays = st["rays_closest"] + st["rays_shadow"]
    n_wide, why = scene5.wide_records()
    out = {
        "workload": f"config5 scene: 10000 base triangles x 1000 instances (10 M instanced), matte/mirror/glass by instance, env light, "
                    f"PathIntegrator max_depth {DEPTH5}, {W5}x{H5}x{SPP5}spp of the 128 = one pass of the full job (one GPU; mean of 3 frames after one warm-up; not part of `value`)",
        "value": round(rays / st["total_ms"] / 1e3, 1), "unit": "Mrays/s", "ms_per_frame": round(st["total_ms"], 2),
        "best_frame_Mrays_per_s": round(rays / best["total_ms"] / 1e3, 1),
        "rays_per_frame": int(rays), "trace_only_Mrays_per_s": round(rays / st["trace_ms"] / 1e3, 1),
        "trace_launches": int(st["trace_launches"]), "trace_fraction_of_frame": round(st["trace_ms"] / st["total_ms"], 3),
        "kernel": "k_trace_wide<false, 1>" if n_wide >= 0 else f"k_trace<false, 1> (binary records: {why})",
    }
    if n_wide >= 0:
        ctx.set_counting(2)
        ctx.wide_counters(reset=True)
        go()
        wc = ctx.wide_counters(reset=True)
        ctx.set_counting(0)
        rec_rate = (wc["records"] + wc["triangles"]) / (st["trace_ms"] * 1e-3) / 1e9
        out.update({"records_per_ray": round(wc["records"] / rays, 2), "triangles_per_ray": round(wc["triangles"] / rays, 2),
                    "leaf_candidates_per_ray": round(wc["leaf_candidates"] / rays, 2), "n_records": n_wide,
                    "gather": {"achieved": round(rec_rate, 2), "peak": round(peak_rec, 2), "unit": "G records/s", "frac": round(rec_rate / peak_rec, 4)}})
    scene5.close()
    return out


def scene_interior_bytes(scene):
    n = len(scene.nodes) if getattr(scene, "nodes", None) is not None else 0
    return max(1, (n - 1) // 2) * 64


if __name__ == "__main__":
    main()
