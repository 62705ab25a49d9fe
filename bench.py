#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X path-tracing hot path on BASELINE.json's configs 3, 4 and 5.

  python bench.py [--gpus 1] --steps K --warmup W        config 3: 1 M random triangles + env light, PathIntegrator depth 5,
                                                          1920x1080x64 spp on one GPU
  python bench.py --gpus N --steps K --warmup W          config 4: the same scene at 256 spp in total, the 16x16 tiles of the ONE
                                                          frame dealt to the N ranks in Morton order (total work fixed:
                                                          "scaling": "strong"), one RCCL reduce of the W*H*4 film to rank 0 inside
                                                          the timed step. --scaling weak renders 64 x N spp instead.
  python bench.py --config 5 --gpus N ...                config 5: 10 000 base triangles x 1000 instances, matte / mirror / glass,
                                                          depth 16, 3840x2160x128 spp (one GPU: four 32-spp passes), tiles split the same way

Both launch forms work at N > 1: started plainly (`python bench.py --gpus N`, no RANK + WORLD_SIZE in the environment) this process
starts the N ranks itself as CHILD processes — before it has imported torch.cuda or the HIP library, so nothing that has
initialised the GPU is ever replaced — and leaves with their exit code; started under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` (RANK and WORLD_SIZE set) it is one of the ranks.

A step = one frame through Integrator::render (scene + BVH resident in HBM, film left on the device). One ray = one
Scene::intersect / intersect_p call. Prints ONE JSON line on rank 0. A rank that fails or hangs ends the job with a non-zero
exit code (3 = watchdog, 4 = a stage failed on some rank; the line, when there is one, is printed first).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# BASELINE.json's configurations 3, 4, 5; `tris` = base triangles of the instanced scene for config 5
CONFIGS = {
    3: dict(scene="random_triangles", width=1920, height=1080, spp=64, max_depth=5, tris=1_000_000, instances=0, spp_per_pass=0),
    4: dict(scene="random_triangles", width=1920, height=1080, spp=256, max_depth=5, tris=1_000_000, instances=0, spp_per_pass=0),
    # spp_per_pass 0 = as many samples of a pixel in flight as HBM and the 2^28-path bound allow: four 32-spp passes of the whole
    # 4K frame on one GPU, one 128-spp pass of a rank's eighth of the tiles at N = 8 (profiles/r05_rank_of_world.txt)
    5: dict(scene="instanced", width=3840, height=2160, spp=128, max_depth=16, tris=10_000, instances=1000, spp_per_pass=0),
}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_traffic.json")
# one-GPU anchors of the N-GPU curve (config4_n1 of a committed one-GPU line), newest first
ANCHOR_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r05_bench_line.json", "r04_bench_line.json")]
# what a stamped counter profile was measured on: the kernels, and what decides the workload they were given
KERNEL_SOURCES = ("pbrt-rs_amd/csrc", "pbrt-rs_amd/build.sh", "bench.py", "pbrt-rs_amd/pbrt_hip/scenes.py")


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=0,
                    help="BASELINE.json configuration: 3 (default at one GPU), 4 (default at N > 1), 5 (10 M instanced triangles, 4K x 128 spp)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0,
                    help="samples per pixel of the frame (strong) or per GPU (weak); default: the configuration's")
    ap.add_argument("--tris", type=int, default=0, help="triangles (config 5: of the base mesh)")
    ap.add_argument("--instances", type=int, default=0, help="config 5: instances of the base mesh")
    ap.add_argument("--max-depth", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[640, 360], help="crop rendered by the CPU oracle")
    ap.add_argument("--cpu-spp", type=int, default=48)
    ap.add_argument("--spp-per-pass", type=int, default=-1, help="samples of a pixel traced concurrently (default: the configuration's; 0 = sized from free HBM)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default): the tiles of ONE frame are split over the GPUs (BASELINE configs 4 and 5); "
                         "weak: the frame has spp x N samples per pixel (per-GPU work fixed)")
    ap.add_argument("--tile-order", choices=("morton", "row-major"), default="morton",
                    help="order in which the 16x16 tiles are dealt to the ranks (PbrtRenderParams.tile_order; SURVEY 8(e): Morton)")
    ap.add_argument("--abi-reduce-check", action="store_true", help="(default at N > 1 over RCCL; kept for old command lines)")
    ap.add_argument("--no-abi-reduce-check", action="store_true",
                    help="skip the untimed repeat of the film merge through the C ABI's own RCCL communicator (pbrt_hip_comm_create / "
                         "pbrt_hip_film_reduce), which otherwise runs after the measurement at N > 1 and is compared with torch.distributed's")
    ap.add_argument("--abi-check-timeout-s", type=float, default=120.0,
                    help="the ABI reduce check may take this long; after that the measured line is printed with abi_film_reduce = TIMEOUT")
    ap.add_argument("--abi-check-strict", action="store_true",
                    help="a failure of the ABI reduce check ends the job with exit code 4 (default: the measured line is printed with "
                         "config.abi_film_reduce = FAILED ... and a top-level `warnings` entry, exit code 0 — the measurement does not depend on the check)")
    ap.add_argument("--watchdog-s", type=float, default=600.0, help="a rank stuck longer than this ends the job with exit code 3")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl (= RCCL over xGMI, the measured job) or gloo: the same job with the film reduce and the agreement "
                         "collectives staged through host memory — the rehearsal of the N > 1 path on a box with ONE GPU, where "
                         "RCCL refuses two ranks on one device (tests/test_gpu_two_ranks.py)")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses device 0 (rehearsal on a one-GPU box, with --dist-backend gloo)")
    ap.add_argument("--launcher", choices=("spawn", "torchrun"), default="spawn",
                    help="how `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts its ranks: spawn = N child processes of this "
                         "script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (each rank's exit code is kept: 3 watchdog, 4 stage "
                         "failure); torchrun = one child `python -m torch.distributed.run --nproc-per-node N` (its exit code is 1 for any failure)")
    ap.add_argument("--save-film", default="", help="rank 0 writes the reduced film of the last step here (.npy)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed config-5 and config-4 blocks (N = 1, config 3 only)")
    return ap.parse_args(argv)


def resolve_workload(args, world):
    """Fills the workload flags that were not given from the BASELINE configuration (--config; 3 at one GPU, 4 at N > 1) and
    returns (config number, name): "config4" when the run IS that configuration, "config4 variant" when a flag departs from it."""
    cfg = args.config or (3 if world == 1 else 4)
    base = CONFIGS[cfg]
    for key in ("width", "height", "tris", "instances", "max_depth"):
        if getattr(args, key) <= 0:
            setattr(args, key, base[key])
    if args.spp <= 0:
        args.spp = 64 if (args.scaling == "weak" and cfg != 5) else base["spp"]
    if args.spp_per_pass < 0:
        args.spp_per_pass = base["spp_per_pass"]
    spp_total = args.spp * world if args.scaling == "weak" else args.spp
    same = all(getattr(args, k) == base[k] for k in ("width", "height", "tris", "instances", "max_depth")) and spp_total == base["spp"]
    if cfg == 3 and same and world > 1:
        same = False   # config 3 is a one-GPU configuration
    if cfg == 4 and world > 1 and args.scaling != "strong":
        same = False
    return cfg, spp_total, (f"config{cfg}" if same else f"config{cfg} variant")


# ------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without a launcher: the parent of the N ranks. Nothing here may touch the GPU.
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_plan(args, argv, env, port, script=None):
    """What self_launch starts: a list of (command, environment). One entry per rank (spawn) — RANK / LOCAL_RANK / WORLD_SIZE /
    LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them — or the one torch.distributed.run
    command the driver itself uses at N > 1 (torchrun). The ranks get the very same arguments as this process."""
    script = script or os.path.abspath(__file__)
    base = dict(env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    base.setdefault("OMP_NUM_THREADS", "1")              # as torch.distributed.run does for its workers
    base["PBRT_BENCH_PARENT"] = str(os.getpid())
    n = args.gpus
    if args.launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script] + list(argv)
        return [(cmd, base)]
    plan = []
    for rank in range(n):
        e = dict(base)
        e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        plan.append(([sys.executable, script] + list(argv), e))
    return plan


def job_exit_code(codes):
    """The job's exit code from the ranks' (None = a rank this parent had to end): rank 0's when it failed, else the first
    failing rank's; a rank that had to be ended counts as a hang (3) unless some rank gave a reason of its own."""
    own = [c for c in codes if c not in (0, None)]
    if not own and any(c is None for c in codes):
        return 3
    if not own:
        return 0
    return codes[0] if codes[0] not in (0, None) else own[0]


def self_launch(args, argv, grace_s=30.0):
    """Starts the ranks as child processes (never an exec: see the module docstring), lets them write to this process's
    stdout / stderr (rank 0 prints the one JSON line), and returns the job's exit code. When a rank leaves non-zero the others
    get `grace_s` to leave by themselves (they do: agreement collectives, broken connections), then are ended by pid."""
    plan = launch_plan(args, argv, os.environ, free_port())

    def die_with_parent():   # a rank must not outlive this process (driver timeout, Ctrl-C)
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)   # PR_SET_PDEATHSIG
        except Exception:
            pass

    procs = [subprocess.Popen(cmd, env=env, cwd=os.getcwd(), preexec_fn=die_with_parent) for cmd, env in plan]

    def forward(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    for s in (signal.SIGINT, signal.SIGTERM):
        signal.signal(s, forward)

    first_failure = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        if first_failure is None and any(p.poll() not in (None, 0) for p in procs):
            first_failure = time.time()
        if first_failure is not None and time.time() - first_failure > grace_s:
            break
    codes = []
    for p in procs:
        if p.poll() is None:   # still there long after another rank failed: end exactly this pid
            print(f"[bench parent] ending rank process {p.pid}: another rank has failed", file=sys.stderr, flush=True)
            p.terminate()
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            codes.append(None)
        else:
            codes.append(p.returncode)
    return job_exit_code(codes)


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
    """The stamped counter profile of this very command (TRAFFIC_FILE, written by tools/measure_traffic.sh: rocprofv3
    --kernel-trace --stats and separate --pmc passes). It was measured for ONE configuration and ONE state of the kernel
    sources: returned only when this run is that configuration (otherwise None and the reason); `stale` says that the kernel
    sources have changed since (the figures are then reported as measured, flagged)."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except Exception:
        return None, f"{os.path.relpath(TRAFFIC_FILE, ROOT)} missing", False
    mine = dict(n_gpus=world, tris=args.tris, width=args.width, height=args.height, spp=spp_total, max_depth=args.max_depth, kernel=kernel)
    diff = {k: (v, rec["config"].get(k)) for k, v in mine.items() if rec["config"].get(k) != v}
    if getattr(args, "instances", 0):
        diff["instances"] = (args.instances, 0)
    if diff:
        return None, f"measured for another configuration: {diff}", False
    return rec, None, rec.get("source_hash") != kernel_source_hash()


def profile_source_text(rec, why_not, hash_differs, drift):
    """One sentence on where the counter-derived fields come from and whether they still describe this run: the sources the
    profile was stamped with against today's, and (when the profile holds a launch time) this run's launches against its."""
    if not rec:
        return why_not
    text = (f"{os.path.relpath(TRAFFIC_FILE, ROOT)}: rocprofv3 --pmc passes over this command, commit {rec.get('commit')}, "
            f"sources {rec.get('source_hash')}")
    notes = []
    if hash_differs:
        notes.append("the sources have changed since")
    if drift is not None and abs(drift) > 0.05:
        notes.append(f"this run's launches take {drift * 100:+.1f} % against the profiled ones")
    return text + (" (STALE: " + "; ".join(notes) + ")" if notes else "")


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


def device_identity(torch, device):
    """Which physical device a rank runs on: host name + uuid, or + PCI address where the runtime reports no uuid. Fields the
    torch build does not expose stay empty — an identity nobody can state is UNKNOWN, not equal to every other unknown one."""
    props = torch.cuda.get_device_properties(device)
    uuid = str(getattr(props, "uuid", "") or "")
    pci = [getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    pci_txt = "" if (any(v is None for v in pci) or not any(pci)) else f"{pci[0]:04x}:{pci[1]:02x}:{pci[2]:02x}"
    return {"host": socket.gethostname(), "device": props.name, "uuid": uuid, "pci_bus_id": pci_txt}


def rank_report(torch, device, local_rank, rank, stats, elapsed_s):
    """What one rank of the N-GPU job knows about itself (gathered into config.ranks): which device it really ran on, how
    long ITS renders took (HIP events around the whole render call, per step) and how many rays it traced — load balance
    is what bounds the N-GPU rate, and a line built on N ranks that shared one device would not be a scaling number."""
    free_b, total_b = torch.cuda.mem_get_info(device)
    ms = [st["total_ms"] for st in stats] or [0.0]
    rep = {"rank": rank, "local_rank": local_rank}
    rep.update(device_identity(torch, device))
    rep.update({
        "hbm_free_GB": round(free_b / 1e9, 1), "hbm_total_GB": round(total_b / 1e9, 1),
        "render_ms_per_step": {"mean": round(sum(ms) / len(ms), 2), "max": round(max(ms), 2)},
        "trace_ms_per_step": round(sum(st["trace_ms"] for st in stats) / max(len(stats), 1), 2),
        "rays_per_step": int(sum(st["rays_closest"] + st["rays_shadow"] for st in stats) / max(len(stats), 1)),
        "wall_s": round(elapsed_s, 4), "runtime_libs": loaded_runtime_libs(), "pid": os.getpid(),
    })
    return rep


def gather_rank_reports(dist, use_dist, world, report):
    """Collective (all_gather_object): every rank's report, in rank order, on every rank."""
    if not use_dist:
        return [report]
    reports = [None] * world
    dist.all_gather_object(reports, report)
    return reports


def check_distinct_devices(reports, one_gpu):
    """N ranks must be N devices unless the job was started as the one-GPU rehearsal (--one-gpu). A device is (host, uuid) —
    or (host, PCI address) where the runtime reports no uuid; a rank whose runtime reports neither is counted as a device of
    its own (unknown is not a collision). Returns the number of distinct devices."""
    known, unknown = [], 0
    for r in reports:
        ident = r.get("uuid") or r.get("pci_bus_id")
        if ident:
            known.append((r.get("host", ""), ident))
        else:
            unknown += 1
    if len(set(known)) != len(known) and not one_gpu:
        raise RuntimeError(f"{len(reports)} ranks on {len(set(known)) + unknown} distinct device(s): {known} "
                           f"(use --one-gpu for the one-GPU rehearsal)")
    return len(set(known)) + unknown


def load_balance(reports):
    """max / mean of the ranks' own render time per step: 1.0 = perfectly even; the N-GPU rate is bounded by 1 / this."""
    ms = [r["render_ms_per_step"]["mean"] for r in reports]
    mean = sum(ms) / len(ms)
    return round(max(ms) / mean, 4) if mean > 0 else None


def check_film_weights(weights, spp):
    """A size-independent property of the merged film (0.5 box filter: FilmTile::add_sample film.rs:252-295 gives every sample
    weight 1 in the pixel it falls into): the filter_weight_sum channel of the frame rank 0 holds after the reduce must be
    `spp` in every pixel and W x H x spp in total — every tile rendered by exactly one rank, every rank's share arrived. The
    slack is one-sided and small: p_film = pixel + u is rounded to float32, so a sample within half an ulp of a pixel border
    (ulp(1900) = 1.2e-4: about one sample in 10^4 at 1080p, one in 4000 at 4K) lies ON the border and add_sample's footprint
    p0 = ceil(x - 1), p1 = floor(x) + 1 gives it to both neighbours (the oracle does the same: the films are equal bit for bit).
    So the total may exceed W x H x spp by up to 10^-3 of it, never fall short, and no pixel strays by more than 2 + spp / 32."""
    h, w = weights.shape
    total, expected = float(weights.sum()), float(w) * h * spp
    off = int((np.abs(weights - spp) > 2.0 + spp / 32.0).sum())
    return {"pixels": int(w * h), "spp": int(spp), "weight_sum": total, "expected": expected, "excess": total - expected,
            "pixels_off": off, "min": float(weights.min()), "max": float(weights.max()),
            "ok": bool(off == 0 and 0.0 <= total - expected <= 1e-3 * expected)}


def scaling_anchor(config_name):
    """The one-GPU point of the workload the N-GPU job runs (config 4: `config4_n1` of a committed one-GPU bench line —
    another run, possibly another box: a diagnostic beside the driver's own curve, never `vs_baseline`)."""
    if config_name != "config4":
        return None
    for path in ANCHOR_FILES:
        try:
            line = json.load(open(path))
            v = (line.get("config4_n1") or {}).get("value")
            if v:
                return {"value": v, "unit": "Mrays/s", "source": os.path.relpath(path, ROOT) + ": config4_n1 (one GPU, the same 256-spp frame)"}
        except Exception:
            continue
    return None


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


def roofline_block(kernel, trace_s_per_launch, launches_per_frame, rays_per_launch, alg_bytes_per_ray, alg_counts, gather,
                   compulsory_per_launch, rec, why_not, hash_differs, trace_fraction_of_step):
    """The `roofline` object of the bench line, under BASELINE.md section 4 / SURVEY.md 8(d): the bound is HBM, `achieved` is
    the dominant kernel's fabric-side bytes per launch (FETCH_SIZE + WRITE_SIZE of the stamped rocprofv3 counter profile —
    counters cannot be read inside a run) over THIS run's launch time (HIP events on the library's stream), `peak` 8 TB/s,
    `frac` = achieved / peak: a fraction of bytes really moved, hence <= 1 up to Infinity-Cache hits the counter includes.
    `algorithmic_frac` is 8(d)'s own figure — the bytes the REFERENCE's loop touches for these rays (32 B per box test, 48 B
    per triangle test, counted in this run by the instrumented kernel) over the same time and peak; it exceeds 1 because the
    kernel does not move those bytes (4-wide 48-B records, cache-resident tree): reported, not a fraction of anything physical.
    `ta_busy`, `valu_issue` (stamped) and `gather_frac` (this run) are the diagnostics of the units the kernel really waits
    for. `measured_in_this_run` says per field which of the two sources it has."""
    tr = rec["trace"] if rec else None
    stamped_launch_ms = tr["avg_launch_ns_under_kernel_trace"] * 1e-6 if tr and tr.get("avg_launch_ns_under_kernel_trace") else None
    drift = (trace_s_per_launch * 1e3 / stamped_launch_ms - 1.0) if (stamped_launch_ms and trace_s_per_launch > 0) else None
    # the stamped counters belong to the launches they were collected on: a run whose own launches take a different time
    # (another box, clock, driver) is not described by them either
    stale = bool(hash_differs or (drift is not None and abs(drift) > 0.05)) if rec else None
    traffic = tr["bytes_per_launch"] if tr and tr.get("bytes_per_launch") else None
    achieved = traffic / trace_s_per_launch / 1e9 if (traffic and trace_s_per_launch > 0) else None
    peak_ips = 256 * 4 * 2.4e9 / 2   # one wave64 vector instruction per two cycles per SIMD, 1024 SIMDs at 2.4 GHz
    valu = tr["valu_insts_per_launch"] / trace_s_per_launch / peak_ips if (tr and tr.get("valu_insts_per_launch") and trace_s_per_launch > 0) else None
    alg_per_launch = alg_bytes_per_ray * rays_per_launch
    alg_gbps = alg_per_launch / trace_s_per_launch / 1e9 if trace_s_per_launch > 0 else None

    def r4(v):
        return round(v, 4) if v is not None else None
    block = {
        "kernel": kernel, "bound": "HBM", "unit": "GB/s", "peak": HBM_PEAK_GBS,
        "achieved": round(achieved, 1) if achieved is not None else None,
        "frac": r4(achieved / HBM_PEAK_GBS) if achieved is not None else None,
        "traffic": traffic,
        "definition": "BASELINE.md section 4: HBM-bound. achieved = (FETCH_SIZE + WRITE_SIZE) per launch of the dominant kernel, stamped "
                      "rocprofv3 counters, / this run's average launch time; frac = achieved / 8 TB/s (bytes really moved; Infinity-Cache "
                      "hits are in the counter). algorithmic_frac = SURVEY 8(d)'s bytes of the REFERENCE's loop for these rays / the same "
                      "time / 8 TB/s: above 1, because the kernel walks 4-wide 48-B records of a cache-resident tree instead",
        "algorithmic_frac": r4(alg_gbps / HBM_PEAK_GBS) if alg_gbps is not None else None,
        "ta_busy": r4(tr.get("ta_busy_fraction")) if tr else None,
        "valu_issue": r4(valu),
        "gather_frac": r4(gather["achieved"] / gather["peak"]) if gather and gather.get("peak") else None,
        "valu_lane_utilisation": round(tr["valu_lane_utilisation"], 3) if tr and tr.get("valu_lane_utilisation") else None,
        "waves_waiting_fraction": round(tr["wave_cycles_waiting_fraction"], 3) if tr and tr.get("wave_cycles_waiting_fraction") else None,
        "measured_in_this_run": {"avg_launch_ms": True, "algorithmic_frac": True, "gather_frac": True, "achieved": False, "frac": False,
                                 "traffic": False, "ta_busy": False, "valu_issue": False, "valu_lane_utilisation": False,
                                 "waves_waiting_fraction": False},
        "counter_derived": None if rec else f"null: {why_not} (achieved, frac, traffic, ta_busy, valu_issue); algorithmic_frac / gather / wide are this run's own, rank 0's tile share",
        "stamped_avg_launch_ms": round(stamped_launch_ms, 4) if stamped_launch_ms else None,
        "launch_time_vs_stamped": r4(drift), "stale": stale, "source": profile_source_text(rec, why_not, hash_differs, drift),
        "launches_per_step": launches_per_frame, "avg_launch_ms": round(trace_s_per_launch * 1e3, 4),
        "rays_per_launch": round(rays_per_launch), "trace_fraction_of_step": trace_fraction_of_step,
        "hbm": {"compulsory_bytes_per_launch": round(compulsory_per_launch),
                "compulsory_GBps": round(compulsory_per_launch / trace_s_per_launch / 1e9, 1) if trace_s_per_launch > 0 else None,
                "counter_over_compulsory": round(traffic / compulsory_per_launch, 2) if traffic and compulsory_per_launch else None},
        # SURVEY 8(d)'s figure: the bytes the REFERENCE's loop touches for these rays
        "algorithmic": dict({"bytes_per_launch": round(alg_per_launch), "GBps": round(alg_gbps, 1) if alg_gbps is not None else None,
                             "bytes_per_ray": round(alg_bytes_per_ray, 1),
                             "note": "served mostly from L2 / Infinity Cache: not a fraction of HBM peak"}, **alg_counts),
        # the kernel's own fetches (48-B records + triangles, counted by k_trace_wide<COUNT>) against dependent record
        # fetches from an L1-resident table, both measured in this run (pbrt_hip_probe_gather)
        "gather": gather,
    }
    return block


def shade_block(rec, stale):
    """roofline.shade: k_shade streams the SoA path state (stamped profile; its launch time is the profile's own, the library
    times only the traversal launches). FETCH_SIZE under-counts the coalesced streams it reads (counted at 1/2 per line) and
    counts the triangle gathers per line touched: against the bytes MOVED it reads 0.62 at shade-queue density 0.7 and 0.633 at
    density 1.0 (profiles/shade_fetch_calibration.json). A step's launches run from density 1.0 on the first bounce downwards,
    so the figure is given as a RANGE over the two factors; WRITE_SIZE is the traffic at face value."""
    sh = rec.get("shade") if rec else None
    if not (sh and sh.get("avg_launch_ns_under_kernel_trace")):
        return None
    cal = sh.get("fetch_calibration") or {}
    f_sparse, f_dense = cal.get("factor"), cal.get("factor_dense") or cal.get("factor")
    ns = sh["avg_launch_ns_under_kernel_trace"]

    def gbps(factor):
        return (sh["fetch_bytes_per_launch"] / factor + sh["write_bytes_per_launch"]) / ns
    face = sh["bytes_per_launch"] / ns
    hi = gbps(f_sparse) if f_sparse else face
    lo = gbps(f_dense) if f_sparse else face
    return {"kernel": "k_shade", "bound": "HBM", "achieved": round(hi, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(hi / HBM_PEAK_GBS, 4), "frac_range": [round(lo / HBM_PEAK_GBS, 4), round(hi / HBM_PEAK_GBS, 4)],
            "traffic": round(hi * ns), "traffic_counters_at_face_value": sh["bytes_per_launch"],
            "frac_counters_at_face_value": round(face / HBM_PEAK_GBS, 4),
            "source": (f"FETCH_SIZE / f + WRITE_SIZE, f = FETCH_SIZE over the bytes MOVED on k_shade's access pattern: {f_sparse} at shade-queue "
                       f"density {cal.get('density', 0.7)} ... {f_dense} at density {cal.get('density_dense', 1.0)} ({cal.get('source')}); "
                       f"a step's launches run from density 1.0 downwards, so the fraction lies in frac_range"
                       if f_sparse else "counters at face value (uncalibrated)"),
            "measured_in_this_run": False, "avg_launch_ms": round(ns * 1e-6, 4), "launches_per_step": sh.get("launches_per_step"),
            "valu_lane_utilisation": sh.get("valu_lane_utilisation"), "stale": stale}


def build_scene(args, cfg, pbrt_hip, scenes, local_rank):
    """Scene + camera of the configuration, BVH built on the host (timed), resident in HBM behind a fresh context."""
    W, H = args.width, args.height
    t0 = time.time()
    if CONFIGS[cfg]["scene"] == "instanced":
        sc = scenes.instanced_scene(args.tris, args.instances)
        cam = scenes.instanced_camera(W, H)
        bvh = pbrt_hip.build_two_level(sc)
    else:
        sc = scenes.random_triangles(args.tris, seq=1)
        cam = scenes.random_triangles_camera(W, H)
        t0 = time.time()
        bvh = pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH)
    t_bvh = time.time() - t0
    ctx = pbrt_hip.Context(local_rank)
    scene = pbrt_hip.Scene(ctx, sc, bvh=bvh)
    return sc, cam, t_bvh, ctx, scene


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and not ("WORLD_SIZE" in os.environ and "RANK" in os.environ):
        # no launcher (a launcher sets RANK and WORLD_SIZE): be the parent of the ranks. Nothing has touched the GPU in this
        # process (torch is not even imported).
        sys.exit(self_launch(args, argv))
    run_rank(args)


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    import torch
    import torch.distributed as dist
    import pbrt_hip
    from pbrt_hip import scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible (the HIP path has no CPU fallback)")
    if args.one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        # a node with fewer visible devices than ranks: say so and leave with the stage-failure code BEFORE the rendezvous, so
        # that the launcher (or the parent above) ends the job in seconds instead of the peers waiting for this rank
        print(f"[bench rank {rank}] LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
              f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): --gpus {args.gpus} needs one device per rank", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(4)
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
    # rank with exit code 3; the launcher (or the parent above) then ends the other ranks.
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

    cfg, spp_total, config_name = resolve_workload(args, world)
    W, H = args.width, args.height
    tile_order = 0 if args.tile_order == "morton" else 1
    rc = 0
    try:
        # which devices the ranks are on, BEFORE the scene is built and the steps are timed: a mis-launched job (two ranks on
        # one device) stops here, not after the whole run
        ident = stage.run("device identity", lambda: device_identity(torch, device))
        idents = stage.collective("gather device ids", lambda: gather_rank_reports(dist, use_dist, world, ident))
        n_devices = stage.run("distinct devices", lambda: check_distinct_devices(idents, args.one_gpu))

        sc, cam, t_bvh, ctx, scene = stage.run("scene", lambda: build_scene(args, cfg, pbrt_hip, scenes, local_rank))
        n_wide, wide_reason = scene.wide_records()
        inst = 1 if CONFIGS[cfg]["scene"] == "instanced" else 0
        kernel = "k_trace_wide" if n_wide >= 0 else "k_trace"
        kernel_label = f"{kernel}<false, {inst}>"
        # Two films: the reduce of frame k (torch's stream) may still be reading its film while frame k+1 is rendered
        # (the library's stream)
        films = [torch.zeros((H, W, 4), dtype=torch.float32, device=device) for _ in range(2 if use_dist else 1)]

        def render_into(film, spp=None, **kw):
            return scene.render(cam, W, H, spp or spp_total, max_depth=args.max_depth, rr_threshold=1.0, light_strategy=1, seed=0,
                                tile_rank=rank, tile_world=world, tile_order=tile_order, spp_per_pass=args.spp_per_pass,
                                d_film_ptr=film.data_ptr(), **kw)[1]

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

        film_check = [None]

        def timed():
            seconds, stats, last = run_steps(render_into, films, args.steps, args.warmup, dist, use_dist, torch.cuda.synchronize, record_event, stage,
                                             reduce_staged if (staged and use_dist) else None)
            if args.save_film and rank == 0:
                np.save(args.save_film, last.cpu().numpy())
            if rank == 0:   # untimed: the merged film of the last step holds every sample of the frame exactly once
                film_check[0] = check_film_weights(last[..., 3].double().cpu().numpy(), spp_total)
            return (seconds, sum(st["rays_closest"] + st["rays_shadow"] for st in stats), sum(st["trace_ms"] for st in stats),
                    sum(st["trace_launches"] for st in stats), stats)

        elapsed, rays, trace_ms, trace_launches, step_stats = timed()   # carries its own agreement points (run_steps)
        my_rays, my_elapsed = rays, elapsed
        # every rank's own account of the run (device identity, its render times, its rays): config.ranks
        report = stage.run("rank report", lambda: rank_report(torch, device, local_rank, rank, step_stats, elapsed))
        reports = stage.collective("gather rank reports", lambda: gather_rank_reports(dist, use_dist, world, report))
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
            # The measured line first: value, timing, configuration, the ranks' reports. What follows it — instrumented renders, gather
            # probes, the config-5 / config-4 side blocks, the CPU oracle — is untimed diagnostics: a failure THERE is reported in the
            # line (`warnings`, roofline.error) and cannot take the measurement with it.
            split = "one GPU" if world == 1 else (f"{args.spp} spp per GPU" if args.scaling == "weak" else "tiles of the one frame split over the GPUs")
            scene_txt = (f"{args.tris} base triangles x {args.instances} instances, matte / mirror / glass by instance, constant env light"
                         if inst else f"{args.tris} random triangles + constant env light")
            out.update({
                "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
                "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {
                    "workload": f"{config_name}: {scene_txt}, PathIntegrator max_depth {args.max_depth}, {W}x{H}x{spp_total}spp ({split}), "
                                f"SAH BVH <=4 prims/leaf, seed 0",
                    "parallelism": (f"tiles16x16 dealt in {args.tile_order} order to {world} rank(s); "
                                    + ("film reduce staged through host memory (gloo), all ranks on GPU 0: a rehearsal, not a scaling number"
                                       if staged else "RCCL film reduce") + " inside the step") if world > 1 else "1 GPU",
                    "sec_per_frame": round(elapsed / args.steps, 4),
                    "rays_per_frame": int(rays / args.steps),
                    "bvh_build_s_host": round(t_bvh, 2),
                    "traversal": f"{kernel_label}" + (f" ({n_wide} 4-wide records)" if n_wide >= 0 else f" (binary records: {wide_reason})"),
                },
                "roofline": None, "cpu_baseline": None, "secondary": None, "config4_n1": None,
            })
            out["config"]["dist_backend"] = args.dist_backend if use_dist else None
            out["config"]["launcher"] = ("bench.py --gpus N (its own child processes)" if os.environ.get("PBRT_BENCH_PARENT") and "TORCHELASTIC_RUN_ID" not in os.environ
                                         else ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else ("environment" if use_dist else None)))
            out["config"]["tile_order"] = args.tile_order
            out["config"]["runtime_libs"] = loaded_runtime_libs()
            # per rank: device identity, its own render time per step, its rays (DESIGN.md section 6 says how to read them)
            out["config"]["ranks"] = reports
            out["config"]["n_devices"] = n_devices
            out["config"]["load_balance_max_over_mean"] = load_balance(reports)
            out["config"]["film_check"] = film_check[0]
            if film_check[0] and not film_check[0]["ok"]:
                out.setdefault("warnings", []).append("film_check failed: the merged film does not hold every sample once")
            anchor = scaling_anchor(config_name) if world > 1 else None
            if anchor:
                out["scaling_vs_config4_n1"] = dict(anchor, ratio=round(value / anchor["value"], 3),
                                                    note="this job's rate over the one-GPU rate of the same frame from ANOTHER run (see source): a reading aid; "
                                                         "the driver computes the curve from its own per-N runs")

            def diagnostics():
                stage.inject("diagnostics")   # test hook (PBRT_BENCH_FAIL=diagnostics@0)
                launches_per_frame = trace_launches / args.steps
                trace_s_per_launch = trace_ms * 1e-3 / max(trace_launches, 1)
                rays_per_launch = my_rays / max(trace_launches, 1)
                # ---- untimed, instrumented renders of this rank's tile set (per-ray figures do not depend on spp: one pass of the
                # frame is enough where the frame has several) ----
                instr_spp = min(spp_total, args.spp_per_pass) if args.spp_per_pass > 0 else min(spp_total, 256 if not inst else 32)
                # (1) the reference's loops: box / triangle tests of BVHAccel::intersect for exactly these rays -> SURVEY 8(d)
                ctx.set_counting(1)
                ctx.counters(reset=True)
                st_c = render_into(films[0], spp=instr_spp)
                c = ctx.counters(reset=True)
                frame_rays = st_c["rays_closest"] + st_c["rays_shadow"]
                alg_bytes = algorithmic_bytes(st_c["rays_closest"], st_c["rays_shadow"], c["node_tests"], c["prim_tests"])
                # (2) what the kernel itself fetches: 48-B records and 48-B triangles, three 16-B lane requests each
                wc = None
                if n_wide >= 0:
                    ctx.set_counting(2)
                    ctx.wide_counters(reset=True)
                    st_w = render_into(films[0], spp=instr_spp)
                    wc = ctx.wide_counters(reset=True)
                    wide_rays = st_w["rays_closest"] + st_w["rays_shadow"]
                ctx.set_counting(0)
                if wc is not None:
                    rec_per_ray = (wc["records"] + wc["triangles"]) / max(wide_rays, 1)
                    rec_bytes, table_bytes, waves = 48, max(n_wide, 1) * 48, 5
                else:   # binary child-pair records: one 64-B record per two box tests
                    rec_per_ray = (c["node_tests"] / 2 + c["prim_tests"]) / max(frame_rays, 1)
                    rec_bytes, table_bytes, waves = 64, scene_interior_bytes(scene), 6
                rec_per_launch = rec_per_ray * rays_per_launch
                achieved_rec = rec_per_launch / trace_s_per_launch / 1e9
                # ---- measured ceilings of the fetch pattern (dependent random record fetches, nothing else to do) ----
                ceil = {
                    "same_footprint_kernel_occupancy": ctx.probe_gather(table_bytes, rec_bytes, waves) / 1e9,
                    "same_footprint_8_waves": ctx.probe_gather(table_bytes, rec_bytes, 8) / 1e9,
                    "l2_resident_8_waves": ctx.probe_gather(2 << 20, rec_bytes, 8) / 1e9,
                    "l1_resident_8_waves": ctx.probe_gather(16 << 10, rec_bytes, 8) / 1e9,
                }
                peak_rec = ceil["l1_resident_8_waves"]
                gather = {"achieved": round(achieved_rec, 2), "peak": round(peak_rec, 2), "unit": "G records/s",
                          "frac": round(achieved_rec / peak_rec, 4), "record_bytes": rec_bytes, "records_per_launch": round(rec_per_launch),
                          "table_bytes": table_bytes, "ceilings_G_records_per_s": {k: round(v, 2) for k, v in ceil.items()}}
                # ---- the stamped counter profile of this command (fabric bytes, issue, texture addressers) ----
                tri_bytes = len(sc["indices"]) * 48
                closest_share = st_c["rays_closest"] / max(frame_rays, 1)
                compulsory_per_ray = 32 + 16 * closest_share + 4 * (1 - closest_share) + 4                       # ray in, hit out, queue entry
                compulsory_per_launch = compulsory_per_ray * rays_per_launch + (table_bytes + tri_bytes)       # + the tree, once
                rec, why_not, hash_differs = measured_traffic(args, world, spp_total, kernel)
                roofline = roofline_block(
                    kernel_label, trace_s_per_launch, launches_per_frame, rays_per_launch, alg_bytes / max(frame_rays, 1),
                    {"node_tests_per_ray": round(c["node_tests"] / max(c["rays"], 1), 2), "tri_tests_per_ray": round(c["prim_tests"] / max(c["rays"], 1), 2)},
                    gather, compulsory_per_launch, rec, why_not, hash_differs,
                    round(trace_ms * 1e-3 / args.steps / (my_elapsed / args.steps), 3))
                shade = shade_block(rec, roofline["stale"])
                if shade:
                    roofline["shade"] = shade
                if wc is not None:
                    roofline["wide"] = {"records_per_ray": round(wc["records"] / max(wide_rays, 1), 2),
                                        "leaf_candidates_per_ray": round(wc["leaf_candidates"] / max(wide_rays, 1), 2),
                                        "triangles_per_ray": round(wc["triangles"] / max(wide_rays, 1), 2),
                                        "rays_left_to_binary_kernel": wc["special_rays"], "n_records": n_wide}
                secondary = None
                config4_n1 = None
                if world == 1 and config_name == "config3" and not args.no_secondary:
                    try:   # reported beside the measurement, never instead of it
                        secondary = secondary_config5(torch, pbrt_hip, scenes, ctx, device, peak_rec)
                    except Exception as e:  # noqa: BLE001
                        secondary = {"error": f"{type(e).__name__}: {e}"}
                    # The N > 1 job is BASELINE config 4 (256 spp, the tiles of the one frame split over the ranks): its
                    # one-GPU point, so that the 1 / 2 / 4 / 8 curve has an anchor on the same workload. Untimed for `value`.
                    try:
                        runs4 = [scene.render(cam, W, H, 256, max_depth=args.max_depth, rr_threshold=1.0, light_strategy=1, seed=0,
                                              tile_order=tile_order, d_film_ptr=films[0].data_ptr())[1] for _ in range(2)]
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
                    cw, ch = min(args.cpu_crop[0], W), min(args.cpu_crop[1], H)
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
                return roofline, cpu_baseline, secondary, config4_n1

            try:
                out["roofline"], out["cpu_baseline"], out["secondary"], out["config4_n1"] = diagnostics()
            except Exception as e:  # noqa: BLE001 - the measurement stands
                try:
                    ctx.set_counting(0)
                except Exception:  # noqa: BLE001
                    pass
                out["roofline"] = {"kernel": kernel_label, "bound": "HBM", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": None, "frac": None,
                                   "traffic": None, "error": f"{type(e).__name__}: {e}"}
                out.setdefault("warnings", []).append(f"untimed diagnostics failed ({type(e).__name__}: {e}); value and timing are unaffected")
                print(f"[bench rank {rank}] untimed diagnostics failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)

        want_abi_check = args.abi_reduce_check or (world > 1 and not args.no_abi_reduce_check)
        if want_abi_check and use_dist and not staged:
            # Untimed, after the measured line is assembled: the same film merge through the C ABI's own RCCL communicator
            # (pbrt_hip_comm_create / pbrt_hip_film_reduce), checked against torch.distributed's reduce. The measurement does
            # not depend on it: a failure or a timeout of the CHECK goes into config.abi_film_reduce and the line is printed.
            stage.collective("before the abi check", barrier)   # rank 0 has just spent seconds on its instrumented renders: the check's clock starts together
            abi_check(args, torch, dist, pbrt_hip, stage, ctx, render_into, device, rank, world, W, H, out, emit)
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


def abi_check(args, torch, dist, pbrt_hip, stage, ctx, render_into, device, rank, world, W, H, out, emit):
    """The film merge once more through pbrt_hip_comm_create / pbrt_hip_film_reduce (csrc/film_reduce.cpp), compared with
    torch.distributed's reduce of the same films. A second communicator that hangs must not take the measured line with it:
    after --abi-check-timeout-s every rank prints / leaves on its own (exit code 0: the measurement is valid, the check says TIMEOUT)."""
    def timed_out():
        if rank == 0 and out:
            out["config"]["abi_film_reduce"] = f"TIMEOUT after {args.abi_check_timeout_s:.0f} s (the check, not the measurement)"
            out.setdefault("warnings", []).append("abi_film_reduce check timed out")
        emit()
        sys.stdout.flush()
        os._exit(4 if args.abi_check_strict else 0)
    timer = threading.Timer(args.abi_check_timeout_s, timed_out)
    timer.daemon = True
    timer.start()
    status = None
    try:
        f = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
        stage.run("abi: render", lambda: render_into(f, spp=max(1, min(args.spp_per_pass, 8) if args.spp_per_pass > 0 else 8)))
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
    except StageFailed as e:
        status = f"FAILED at stage '{e.args[0]}'" + (f": {e.args[1]}" if e.args[1] else " (another rank)")
        print(f"[bench rank {rank}] abi reduce check {status}", file=sys.stderr, flush=True)
        if rank == 0 and out:
            out["config"]["abi_film_reduce"] = status
            out.setdefault("warnings", []).append("abi_film_reduce check " + status)
        emit()
        if args.abi_check_strict:
            raise
        # the ranks can no longer be assumed to be in step (a collective failed somewhere): each leaves on its own
        sys.stdout.flush()
        os._exit(0)
    finally:
        timer.cancel()
    if rank == 0 and out:
        out["config"]["abi_film_reduce"] = status


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
