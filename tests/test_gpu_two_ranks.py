"""The N > 1 job with the HIP kernels in it, in two processes, on the one GPU of the test box.

bench.py --gpus 2 is BASELINE config 4's job: the 16x16 tiles of ONE frame dealt round-robin to the ranks
(parallel.rs:4-21, integrator.rs:412-477), scene replicated, one film reduce to rank 0 inside the step. RCCL refuses two
ranks on one device, so the rehearsal uses --dist-backend gloo (films staged through host memory) and --one-gpu; the
render leg of every rank is pbrt_hip_render_device, exactly as in the measured job. One functional run each, no loops."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, TRIS, DEPTH = 256, 144, 8, 20000, 5
ARGS = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--dist-backend", "gloo", "--one-gpu", "--width", str(W), "--height", str(H),
        "--spp", str(SPP), "--tris", str(TRIS), "--max-depth", str(DEPTH), "--spp-per-pass", str(SPP), "--no-cpu-baseline",
        "--watchdog-s", "240"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("PBRT_BENCH_FAIL", None)
    return env


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_reduce_to_the_one_process_frame(hip_ctx, tmp_path):
    """Launched as the driver launches it (torch.distributed.run, one fresh process per rank): the reduced film is the
    one-process frame bit for bit, and the rays of the two ranks add up to the frame's."""
    film_path = str(tmp_path / "film.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + ARGS + ["--save-film", film_path]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["dist_backend"] == "gloo" and "error" not in line
    assert any("libpbrt_hip" in p for p in line["config"]["runtime_libs"])   # the HIP path is what rendered
    # what makes the first real N-GPU line diagnosable: every rank's own account of the run ...
    ranks = line["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and len({r["pid"] for r in ranks}) == 2
    for r in ranks:
        assert r["device"] and (r["uuid"] or r["pci_bus_id"]) and r["hbm_total_GB"] > 100 and 0 < r["hbm_free_GB"] <= r["hbm_total_GB"]
        assert 0 < r["render_ms_per_step"]["mean"] <= r["render_ms_per_step"]["max"] and r["trace_ms_per_step"] > 0
        assert any("libpbrt_hip" in p for p in r["runtime_libs"])
    # ... the two ranks share the one device of this box, which the line says and --one-gpu permits ...
    assert line["config"]["n_devices"] == 1 and len({(r["host"], r["uuid"] or r["pci_bus_id"]) for r in ranks}) == 1
    assert line["config"]["load_balance_max_over_mean"] >= 1.0
    # ... and a roofline block whose run-measured parts are filled in; only the counter-derived part is null, and says why
    roof = line["roofline"]
    assert roof["bound"] == "HBM" and roof["achieved"] is None and roof["frac"] is None and "null:" in roof["counter_derived"]
    assert roof["measured_in_this_run"]["gather_frac"] is True and roof["measured_in_this_run"]["achieved"] is False
    assert roof["gather"]["achieved"] > 0 and roof["gather_frac"] > 0 and roof["algorithmic_frac"] > 0
    assert roof["algorithmic"]["node_tests_per_ray"] > 1 and roof["avg_launch_ms"] > 0 and roof["wide"]["records_per_ray"] > 1
    assert line["config"]["launcher"] == "torch.distributed.run" and line["config"]["tile_order"] == "morton"
    fc = line["config"]["film_check"]   # the merged film holds every sample of the frame once: W x H x spp in the weight channel
    assert fc["ok"] and fc["pixels"] == W * H and 0 <= fc["excess"] <= 1e-3 * W * H * SPP and "warnings" not in line
    # the same frame in this process, all tiles on one rank
    sc = scenes.random_triangles(TRIS, seq=1)
    g = pbrt_hip.Scene(hip_ctx, sc, bvh=pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH))
    ref, st = g.render(scenes.random_triangles_camera(W, H), W, H, SPP, max_depth=DEPTH, rr_threshold=1.0, light_strategy=1, seed=0)
    g.close()
    film = np.load(film_path)
    assert film.shape == ref.shape
    # non-owned pixels are zero on every rank: the sum is a pure gather (at this size no pixel collects border samples of two
    # foreign tiles; at full size tile-border pixels may differ by the order of float additions: test_film_properties_at_full_size)
    assert film.tobytes() == ref.tobytes()
    assert line["config"]["rays_per_frame"] == st["rays_closest"] + st["rays_shadow"] == sum(r["rays_per_step"] for r in ranks)


def _one_process_frame(hip_ctx, sc, bvh, cam, w, h, spp, depth, **kw):
    g = pbrt_hip.Scene(hip_ctx, sc, bvh=bvh)
    ref, st = g.render(cam, w, h, spp, max_depth=depth, rr_threshold=1.0, light_strategy=1, seed=0, **kw)
    g.close()
    return ref, st


@pytest.mark.timeout(600)
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(hip_ctx, tmp_path):
    """`python bench.py --gpus 2 ...` exactly as the driver starts the one-GPU job — no torch.distributed.run, no WORLD_SIZE:
    the process becomes the parent of two rank processes (before it touches the GPU), rank 0's one line arrives on its stdout,
    its exit code is the job's. Same bit-identical film, same per-rank reports; row-major dealing for once, so that both
    tile orders have gone through the two-process job."""
    film_path = str(tmp_path / "film.npy")
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + ["--save-film", film_path, "--tile-order", "row-major"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "error" not in line and line["config"]["launcher"].startswith("bench.py --gpus N")
    assert line["config"]["tile_order"] == "row-major" and "row-major" in line["config"]["parallelism"]
    ranks = line["config"]["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1] and len({x["pid"] for x in ranks}) == 2 and os.getpid() not in {x["pid"] for x in ranks}
    assert all(any("libpbrt_hip" in p for p in x["runtime_libs"]) for x in ranks)
    sc = scenes.random_triangles(TRIS, seq=1)
    ref, st = _one_process_frame(hip_ctx, sc, pbrt_hip.bvh_build(sc["positions"], sc["indices"], 4, pbrt_hip.SPLIT_SAH),
                                 scenes.random_triangles_camera(W, H), W, H, SPP, DEPTH)
    assert np.load(film_path).tobytes() == ref.tobytes()
    assert line["config"]["rays_per_frame"] == st["rays_closest"] + st["rays_shadow"] == sum(x["rays_per_step"] for x in ranks)
    # --launcher torchrun: the same parent starts ONE child, `python -m torch.distributed.run --nproc-per-node 2 bench.py ...`
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + ["--launcher", "torchrun"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line_t = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line_t["config"]["launcher"] == "torch.distributed.run" and line_t["config"]["film_check"]["ok"]
    assert line_t["config"]["rays_per_frame"] == line["config"]["rays_per_frame"]
    # a rank that fails: the parent's exit code is the ranks' (4), not a launcher's 1
    env["PBRT_BENCH_FAIL"] = "render@1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 4, r.stderr[-3000:]


@pytest.mark.timeout(600)
def test_config5_job_on_two_ranks_equals_the_one_process_frame(hip_ctx, tmp_path):
    """bench.py --config 5 --gpus 2: BASELINE config 5's job (instanced scene, matte / mirror / glass, depth 16, several passes
    of spp_per_pass samples, the tiles of the one frame dealt to the ranks in Morton order) at reduced size — two-level
    traversal kernel in both ranks, film merged on rank 0 == the one-process frame bit for bit."""
    w, h, spp, tris, inst = 320, 180, 8, 2000, 60
    film_path = str(tmp_path / "film5.npy")
    args = ["--config", "5", "--gpus", "2", "--steps", "1", "--warmup", "1", "--dist-backend", "gloo", "--one-gpu", "--width", str(w),
            "--height", str(h), "--spp", str(spp), "--spp-per-pass", "2", "--tris", str(tris), "--instances", str(inst), "--no-cpu-baseline",
            "--watchdog-s", "240", "--save-film", film_path]
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["config"]["workload"].startswith("config5 variant") and "instances" in line["config"]["workload"]
    assert line["roofline"]["kernel"] == "k_trace_wide<false, 1>" and line["roofline"]["wide"]["records_per_ray"] > 1
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    sc = scenes.instanced_scene(tris, inst)
    ref, st = _one_process_frame(hip_ctx, sc, pbrt_hip.build_two_level(sc), scenes.instanced_camera(w, h), w, h, spp, 16, spp_per_pass=2)
    assert np.load(film_path).tobytes() == ref.tobytes()
    ranks = line["config"]["ranks"]
    assert line["config"]["rays_per_frame"] == st["rays_closest"] + st["rays_shadow"] == sum(x["rays_per_step"] for x in ranks)


@pytest.mark.timeout(600)
def test_two_ranks_on_one_device_are_refused_without_one_gpu():
    """The same job WITHOUT --one-gpu is what a mis-launched 2-GPU run on this box would be (LOCAL_RANK 0 for both): the ranks
    gather their device ids right after device selection — before the scene is built or a step is timed — find one device
    behind two ranks and stop, exit code 4: a line from such a run would not be a scaling number."""
    port = _free_port()
    args = [a for a in ARGS if a != "--one-gpu"]
    procs = []
    for rank in range(2):
        env = _env()
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=540) for p in procs]
    assert [p.returncode for p in procs] == [4, 4], [o[1][-1500:] for o in outs]
    assert "distinct device" in outs[0][1] + outs[1][1]
    assert not any(ln.startswith("{") for o in outs for ln in o[0].splitlines())   # stopped before anything was measured


@pytest.mark.timeout(600)
def test_a_rank_that_fails_its_render_ends_both_ranks_with_code_4(tmp_path):
    """PBRT_BENCH_FAIL=render@1: rank 1 fails in the (checked) warm-up render; the agreement collective behind it lets rank 0
    see that instead of waiting in the film reduce: both leave with exit code 4. Two plain child processes (the launcher's
    environment set by hand) so that each rank's own exit code is visible."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = _env()
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PBRT_BENCH_FAIL="render@1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=540) for p in procs]
    assert [p.returncode for p in procs] == [4, 4], [o[1][-1500:] for o in outs]


@pytest.mark.timeout(600)
def test_a_failure_in_the_untimed_diagnostics_does_not_take_the_line():
    """What follows the timed steps on rank 0 — instrumented renders, gather probes, side blocks, the CPU oracle — is diagnostics:
    when it fails (PBRT_BENCH_FAIL=diagnostics@0) the line still carries value, timing, configuration and the film check, says
    what failed (`warnings`, roofline.error), and the exit code stays 0."""
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PBRT_BENCH_FAIL"] = "diagnostics@0"
    args = ["--steps", "1", "--warmup", "1", "--width", str(W), "--height", str(H), "--spp", str(SPP), "--tris", str(TRIS), "--max-depth", str(DEPTH),
            "--no-cpu-baseline", "--no-secondary", "--watchdog-s", "240"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["config"]["film_check"]["ok"] and line["config"]["rays_per_frame"] > 0
    assert line["roofline"]["bound"] == "HBM" and line["roofline"]["frac"] is None and "injected failure" in line["roofline"]["error"]
    assert any("untimed diagnostics failed" in w for w in line["warnings"]) and "untimed diagnostics failed" in r.stderr
    # ... and without the injection the same command fills the block in
    env.pop("PBRT_BENCH_FAIL")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert r.returncode == 0 and "warnings" not in line and line["roofline"]["gather_frac"] > 0 and "error" not in line["roofline"]


@pytest.mark.timeout(600)
def test_more_ranks_than_visible_devices_ends_quickly_with_code_4():
    """`python bench.py --gpus 2` on this one-GPU box WITHOUT --one-gpu: rank 1 finds no device for itself, says so and leaves with
    code 4 before the rendezvous; the parent gives rank 0 (waiting for its peer) the grace period, ends it by pid and returns 4 —
    in well under a minute, not after the 600-second watchdog."""
    import time
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = [a for a in ARGS if a != "--one-gpu"]
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    assert "only 1 GPU(s) are visible" in r.stderr and time.time() - t0 < 240
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
