"""bench.py's host logic that needs no GPU: the kernel-source stamp and the rules under which the stamped counter profile
(bench.TRAFFIC_FILE) may be used for the roofline block; the roofline block itself; how
`python bench.py --gpus N` starts its ranks (argv, environment, exit code); the device check; the workload table."""
import argparse
import json
import os
import subprocess
import sys
import textwrap

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    d = dict(tris=1_000_000, width=1920, height=1080, max_depth=5)
    d.update(kw)
    return argparse.Namespace(**d)


def test_kernel_source_hash_covers_the_kernel_sources(tmp_path, monkeypatch):
    h = bench.kernel_source_hash()
    assert len(h) == 12 and h == bench.kernel_source_hash()
    # a copy of the sources with one byte changed in a kernel header hashes differently
    import shutil
    root = tmp_path / "repo"
    shutil.copytree(os.path.join(ROOT, "pbrt-rs_amd", "csrc"), root / "pbrt-rs_amd" / "csrc")
    shutil.copy(os.path.join(ROOT, "pbrt-rs_amd", "build.sh"), root / "pbrt-rs_amd" / "build.sh")
    # ... and what decides the workload the kernels are given: bench.py itself and the scene generator
    os.makedirs(root / "pbrt-rs_amd" / "pbrt_hip")
    shutil.copy(os.path.join(ROOT, "pbrt-rs_amd", "pbrt_hip", "scenes.py"), root / "pbrt-rs_amd" / "pbrt_hip" / "scenes.py")
    shutil.copy(os.path.join(ROOT, "bench.py"), root / "bench.py")
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.kernel_source_hash() == h
    with open(root / "pbrt-rs_amd" / "csrc" / "trace_wide.h", "a") as f:
        f.write("\n// changed\n")
    h2 = bench.kernel_source_hash()
    assert h2 != h
    with open(root / "pbrt-rs_amd" / "pbrt_hip" / "scenes.py", "a") as f:
        f.write("\n# changed\n")
    h3 = bench.kernel_source_hash()
    assert h3 not in (h, h2)
    with open(root / "bench.py", "a") as f:
        f.write("\n# changed\n")
    assert bench.kernel_source_hash() not in (h, h2, h3)


def test_stamped_profile_is_used_only_for_its_configuration_and_flags_stale_sources(tmp_path, monkeypatch):
    rec = json.load(open(bench.TRAFFIC_FILE))
    cfg = rec["config"]
    assert {"trace", "shade", "commit", "source_hash"} <= set(rec)
    assert rec["trace"]["bytes_per_launch"] > 0 and 0.0 < rec["trace"]["ta_busy_fraction"] < 1.0
    a = _args(tris=cfg["tris"], width=cfg["width"], height=cfg["height"], max_depth=cfg["max_depth"])
    got, why, stale = bench.measured_traffic(a, cfg["n_gpus"], cfg["spp"], cfg["kernel"])
    assert got is not None and why is None
    assert stale == (rec["source_hash"] != bench.kernel_source_hash())
    # another configuration: not used at all, and the reason is given
    got, why, stale = bench.measured_traffic(_args(tris=20_000), 1, cfg["spp"], cfg["kernel"])
    assert got is None and "another configuration" in why and stale is False
    got, why, _ = bench.measured_traffic(a, 2, 256, cfg["kernel"])
    assert got is None and "n_gpus" in why
    # the same profile stamped for other sources: used, flagged
    other = dict(rec, source_hash="000000000000")
    p = tmp_path / "t.json"
    p.write_text(json.dumps(other))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(p))
    got, why, stale = bench.measured_traffic(a, cfg["n_gpus"], cfg["spp"], cfg["kernel"])
    assert got is not None and stale is True
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(tmp_path / "missing.json"))
    assert bench.measured_traffic(a, 1, 64, "k_trace_wide")[0] is None


def test_runtime_libs_lists_only_the_stacks_of_interest():
    libs = bench.loaded_runtime_libs()
    assert isinstance(libs, list) and all(os.path.basename(p).startswith(("libamdhip64", "librccl", "libhsa-runtime64", "libpbrt_hip")) for p in libs)


def _gather(frac=0.5):
    return {"achieved": 70.0 * frac, "peak": 70.0, "unit": "G records/s", "frac": frac}


def _roof(rec, why=None, hash_differs=False, launch_s=None):
    tr = rec["trace"] if rec else None
    launch_s = launch_s or (tr["avg_launch_ns_under_kernel_trace"] * 1e-9 if tr and tr.get("avg_launch_ns_under_kernel_trace") else 0.030)
    return bench.roofline_block("k_trace_wide<false, 0>", launch_s, 6.0, 107.4e6, 4668.4, {"node_tests_per_ray": 137.9, "tri_tests_per_ray": 4.37},
                                _gather(), 5.4e9, rec, why, hash_differs, 0.8)


def test_roofline_block_is_baseline_section_4s():
    """Top level = the contract's: bound HBM, achieved = counter GB/s, peak 8000, frac = achieved / peak (<= 1: bytes really
    moved); SURVEY 8(d)'s algorithmic figure beside it as `algorithmic_frac` (above 1: cache-served), and the diagnostics of
    the units the kernel waits for — at the top level, where the driver's `parsed` keeps them."""
    rec = json.load(open(bench.TRAFFIC_FILE))
    r = _roof(rec)
    assert r["bound"] == "HBM" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    tr = rec["trace"]
    want = tr["bytes_per_launch"] / (tr["avg_launch_ns_under_kernel_trace"] * 1e-9) / 1e9
    assert r["achieved"] == pytest.approx(want, rel=1e-3) and r["traffic"] == tr["bytes_per_launch"]
    assert r["frac"] == pytest.approx(want / 8000.0, abs=1e-4) and 0.0 < r["frac"] <= 1.0
    assert r["algorithmic_frac"] == pytest.approx(4668.4 * 107.4e6 / (tr["avg_launch_ns_under_kernel_trace"] * 1e-9) / 8e12, rel=1e-3)
    assert r["algorithmic_frac"] > 1.0 and "cache" in r["definition"]
    assert r["ta_busy"] == pytest.approx(tr["ta_busy_fraction"], abs=1e-4) and 0 < r["valu_issue"] < 1 and r["gather_frac"] == 0.5
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_frac", "ta_busy", "valu_issue", "gather_frac",
                "measured_in_this_run", "stale", "source", "avg_launch_ms", "launches_per_step"):
        assert key in r, key
    m = r["measured_in_this_run"]
    assert m["avg_launch_ms"] and m["algorithmic_frac"] and m["gather_frac"]
    assert not (m["achieved"] or m["frac"] or m["traffic"] or m["ta_busy"] or m["valu_issue"])
    assert r["stale"] is False and "STALE" not in r["source"]
    # this run's launches 8 % slower than the profiled ones, or other sources: flagged, and the text says which
    slow = _roof(rec, launch_s=tr["avg_launch_ns_under_kernel_trace"] * 1e-9 * 1.08)
    assert slow["stale"] is True and "+8.0 %" in slow["source"] and "sources have changed" not in slow["source"]
    changed = _roof(rec, hash_differs=True)
    assert changed["stale"] is True and "the sources have changed since" in changed["source"] and "%" not in changed["source"].split("STALE")[1]
    # no profile for this configuration: the counter-derived fields are null and say why, the run's own are filled
    none = _roof(None, why="measured for another configuration: {...}")
    assert none["bound"] == "HBM" and none["achieved"] is None and none["frac"] is None and none["traffic"] is None and none["ta_busy"] is None
    assert "null:" in none["counter_derived"] and none["algorithmic_frac"] > 0 and none["gather_frac"] == 0.5 and none["stale"] is None


def test_a_stale_profile_without_a_stamped_launch_time_still_gives_a_line():
    """a profile that is stale by source hash and holds no avg_launch_ns_under_kernel_trace (the --stats CSV was
    missing, or an older format) must not raise while the roofline block is put together."""
    rec = json.load(open(bench.TRAFFIC_FILE))
    rec = dict(rec, trace=dict(rec["trace"], avg_launch_ns_under_kernel_trace=None))
    r = _roof(rec, hash_differs=True, launch_s=0.031)
    assert r["stale"] is True and r["launch_time_vs_stamped"] is None and r["stamped_avg_launch_ms"] is None
    assert "the sources have changed since" in r["source"] and r["achieved"] > 0
    del rec["trace"]["avg_launch_ns_under_kernel_trace"]
    assert _roof(rec, hash_differs=True, launch_s=0.031)["stale"] is True
    assert "STALE" in bench.profile_source_text(rec, None, True, None) and "STALE" not in bench.profile_source_text(rec, None, False, 0.01)


def test_shade_block_states_its_calibration_as_a_range():
    """the 0.62 factor was measured at shade-queue density 0.7, a step's launches start at density 1.0 (0.70):
    the figure carries that as frac_range, with the face-value fraction beside it."""
    rec = json.load(open(bench.TRAFFIC_FILE))
    sh = bench.shade_block(rec, False)
    lo, hi = sh["frac_range"]
    assert sh["frac_counters_at_face_value"] < lo < hi == sh["frac"] <= 1.0
    assert "density" in sh["source"] and sh["measured_in_this_run"] is False
    assert bench.shade_block(None, None) is None


def test_self_launch_plan_argv_and_environment():
    """`python bench.py --gpus N` without a launcher: N child processes of this very script with the same arguments and
    torch.distributed.run's environment (or, --launcher torchrun, the one command the driver itself uses at N > 1)."""
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "5"]
    args = bench.parse(argv)
    plan = bench.launch_plan(args, argv, {"PATH": "/bin", "OMP_NUM_THREADS": "7"}, 23456, script="/x/bench.py")
    assert len(plan) == 4
    for rank, (cmd, env) in enumerate(plan):
        assert cmd == [sys.executable, "/x/bench.py"] + argv
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "23456"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["OMP_NUM_THREADS"] == "7" and env["PATH"] == "/bin"
        assert env["PBRT_BENCH_PARENT"] == str(os.getpid())
    argv_t = argv + ["--launcher", "torchrun"]
    (cmd, env), = bench.launch_plan(bench.parse(argv_t), argv_t, {}, 23456, script="/x/bench.py")
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "23456", "/x/bench.py"] + argv_t
    assert "RANK" not in env and env["OMP_NUM_THREADS"] == "1"


def test_job_exit_code_keeps_the_ranks_codes():
    assert bench.job_exit_code([0, 0, 0]) == 0
    assert bench.job_exit_code([4, 4]) == 4 and bench.job_exit_code([0, 4]) == 4
    assert bench.job_exit_code([3, 4]) == 3          # rank 0's own reason first
    assert bench.job_exit_code([None, 4]) == 4       # a rank the parent had to end, beside one that said why
    assert bench.job_exit_code([0, None]) == 3       # nobody said why: a hang


def test_self_launch_runs_the_ranks_as_children_and_forwards_line_and_exit_code(tmp_path, monkeypatch, capfd):
    """The parent end of `python bench.py --gpus 2` with a stand-in rank program (no GPU here): both ranks start with their
    environment, rank 0's one line arrives on this process's stdout, the job's exit code is the failing rank's; a rank that
    stays behind after another failed is ended (by pid) after the grace period."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys, time
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        mode = sys.argv[sys.argv.index("--mode") + 1]
        if mode == "hang" and rank == 0:
            time.sleep(600)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "argv": sys.argv[1:], "port": os.environ["MASTER_PORT"]}), flush=True)
        sys.exit({"ok": 0, "fail": 4 if rank == 1 else 0, "hang": 4}[mode])
    """))
    real_plan = bench.launch_plan

    def plan_with_stand_in(args, argv, env, port, script_=None):
        return real_plan(args, argv + ["--mode", mode], env, port, script=str(script))
    monkeypatch.setattr(bench, "launch_plan", plan_with_stand_in)
    for mode, want in (("ok", 0), ("fail", 4), ("hang", 4)):
        argv = ["--gpus", "2", "--steps", "1"]
        t0 = __import__("time").time()
        rc = bench.self_launch(bench.parse(argv), argv, grace_s=1.0)
        out = capfd.readouterr()
        assert rc == want, (mode, rc, out.err[-500:])
        lines = [ln for ln in out.out.splitlines() if ln.startswith("{")]
        if mode != "hang":
            assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and json.loads(lines[0])["argv"][:4] == argv
        else:
            assert not lines and "ending rank process" in out.err and __import__("time").time() - t0 < 60


def test_main_becomes_the_parent_only_without_a_launcher(monkeypatch):
    """--gpus N > 1 and no WORLD_SIZE: self_launch (before torch.cuda / the HIP library are imported); with WORLD_SIZE set
    (torch.distributed.run, or a rank of the parent) the process is a rank; --gpus 1 never launches."""
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda args, argv: calls.append(("parent", args.gpus)) or 0)
    monkeypatch.setattr(bench, "run_rank", lambda args: calls.append(("rank", args.gpus)))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert e.value.code == 0 and calls == [("parent", 8)]
    bench.main(["--gpus", "1"])
    monkeypatch.setenv("WORLD_SIZE", "8")          # a stray WORLD_SIZE without RANK is not a launcher
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "8"])
    monkeypatch.setenv("RANK", "3")
    bench.main(["--gpus", "8"])
    assert calls == [("parent", 8), ("rank", 1), ("parent", 8), ("rank", 8)]
    # the parent path imports neither torch nor the HIP binding: a fresh interpreter that starts ranks which do nothing
    code = ("import sys, bench; sys.modules_before = set(sys.modules); "
            "bench.launch_plan = lambda a, argv, env, port, script=None: [([sys.executable, '-c', 'pass'], dict(env))] * 2; "
            "rc = bench.self_launch(bench.parse(['--gpus', '2']), ['--gpus', '2']); "
            "assert rc == 0 and 'torch' not in sys.modules and 'pbrt_hip' not in sys.modules")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]


def test_workload_table_is_baselines():
    """--config / --gpus pick BASELINE.json's configurations 3, 4, 5; a flag that departs from one is named a variant."""
    a = bench.parse([])
    assert bench.resolve_workload(a, 1) == (3, 64, "config3") and (a.width, a.height, a.tris, a.max_depth, a.spp_per_pass) == (1920, 1080, 1_000_000, 5, 0)
    a = bench.parse(["--gpus", "8"])
    assert bench.resolve_workload(a, 8) == (4, 256, "config4") and a.spp == 256
    a = bench.parse(["--gpus", "8", "--config", "5"])
    assert bench.resolve_workload(a, 8) == (5, 128, "config5")
    assert (a.width, a.height, a.tris, a.instances, a.max_depth, a.spp_per_pass) == (3840, 2160, 10_000, 1000, 16, 0)
    a = bench.parse(["--gpus", "4", "--scaling", "weak"])
    assert bench.resolve_workload(a, 4) == (4, 256, "config4 variant") and a.spp == 64
    a = bench.parse(["--gpus", "2", "--width", "256", "--height", "144"])
    assert bench.resolve_workload(a, 2)[2] == "config4 variant"
    a = bench.parse(["--config", "5", "--tris", "2000", "--instances", "50"])
    assert bench.resolve_workload(a, 1)[2] == "config5 variant"


def test_device_check_unknown_identity_is_not_a_collision():
    """a runtime that reports neither uuid nor PCI address must not make every rank 'the same device'; equal PCI
    addresses on different hosts are different devices; the same device twice is refused unless --one-gpu."""
    unknown = [{"rank": r, "host": "a", "uuid": "", "pci_bus_id": ""} for r in range(4)]
    assert bench.check_distinct_devices(unknown, one_gpu=False) == 4
    two_hosts = [{"host": h, "uuid": "", "pci_bus_id": "0000:05:00"} for h in ("a", "b")]
    assert bench.check_distinct_devices(two_hosts, one_gpu=False) == 2
    same = [{"host": "a", "uuid": "GPU-1", "pci_bus_id": "0000:05:00"}] * 2
    with pytest.raises(RuntimeError, match="distinct device"):
        bench.check_distinct_devices(same, one_gpu=False)
    assert bench.check_distinct_devices(same, one_gpu=True) == 1

    class Props:
        name = "AMD Instinct MI355X"
    class Cuda:
        @staticmethod
        def get_device_properties(_d):
            return Props()
    class Torch:
        cuda = Cuda()
    ident = bench.device_identity(Torch, 0)
    assert ident["uuid"] == "" and ident["pci_bus_id"] == "" and ident["host"]


def test_scaling_anchor_reads_the_committed_one_gpu_line(tmp_path, monkeypatch):
    p = tmp_path / "line.json"
    p.write_text(json.dumps({"config4_n1": {"value": 2850.0}}))
    monkeypatch.setattr(bench, "ANCHOR_FILES", [str(tmp_path / "missing.json"), str(p)])
    a = bench.scaling_anchor("config4")
    assert a["value"] == 2850.0 and "config4_n1" in a["source"]
    assert bench.scaling_anchor("config4 variant") is None and bench.scaling_anchor("config5") is None


def test_film_check_sees_a_missing_and_a_doubled_tile():
    """bench.check_film_weights: the merged film's weight channel is spp everywhere (up to the few samples that fall exactly on a
    pixel border and count in both neighbours); a tile no rank rendered, or one that arrived twice, is seen."""
    import numpy as np
    w = np.full((64, 96), 16.0)
    assert bench.check_film_weights(w, 16)["ok"]
    w[3, 5] += 1.0          # a border sample counted in both neighbours
    ok = bench.check_film_weights(w, 16)
    assert ok["ok"] and ok["max"] == 17.0 and ok["excess"] == 1.0
    missing = w.copy(); missing[16:32, 32:48] = 0.0
    assert not bench.check_film_weights(missing, 16)["ok"] and bench.check_film_weights(missing, 16)["pixels_off"] == 256
    doubled = w.copy(); doubled[0:16, 0:16] *= 2.0
    assert not bench.check_film_weights(doubled, 16)["ok"]
    short = np.full((64, 96), 16.0); short[0, 0] = 15.0    # a lost sample: the total may exceed, never fall short
    assert not bench.check_film_weights(short, 16)["ok"]


def test_ranks_do_not_outlive_the_parent(tmp_path):
    """A driver that gives up on `python bench.py --gpus N` ends the parent: SIGTERM is forwarded to the ranks, and a rank
    whose parent dies without a word (SIGKILL) is ended by the kernel (PR_SET_PDEATHSIG) — no orphan keeps a GPU busy."""
    import signal
    import time
    rank = tmp_path / "rank.py"
    rank.write_text("import os, sys, time\nopen(sys.argv[1] + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    parent_code = ("import sys, bench\n"
                   "real = bench.launch_plan\n"
                   f"bench.launch_plan = lambda a, argv, env, port, script=None: [([sys.executable, {str(rank)!r}, {str(tmp_path / 'pid')!r}], e) for _, e in real(a, argv, env, port)]\n"
                   "sys.exit(bench.self_launch(bench.parse(['--gpus', '2']), ['--gpus', '2']))\n")

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:   # a zombie of our own child process tree still answers kill(0): look at its state
            return open(f"/proc/{pid}/stat").read().split(") ")[1][0] != "Z"
        except OSError:
            return False

    for sig in (signal.SIGTERM, signal.SIGKILL):
        for f in tmp_path.glob("pid*"):
            f.unlink()
        parent = subprocess.Popen([sys.executable, "-c", parent_code], cwd=ROOT)
        deadline = time.time() + 60
        while time.time() < deadline and len(list(tmp_path.glob("pid*"))) < 2:
            time.sleep(0.1)
        pids = [int(f.read_text()) for f in sorted(tmp_path.glob("pid*"))]
        assert len(pids) == 2 and all(alive(p) for p in pids)
        parent.send_signal(sig)
        parent.wait(30)
        deadline = time.time() + 20
        while time.time() < deadline and any(alive(p) for p in pids):
            time.sleep(0.1)
        assert not any(alive(p) for p in pids), (sig, pids)
