"""bench.py's host logic that needs no GPU: the kernel-source stamp and the rules under which the stamped counter profile
(profiles/r03_traffic.json) may be used for the roofline block."""
import argparse
import json
import os

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
