"""include/pbrt_hip.h is plain C: examples/render_box.c compiles as strict C99 against it, links to the in-tree
library and (on a GPU box) renders; without a GPU it stops at context creation with a non-zero status."""
import os
import re
import subprocess

import pytest

import pbrt_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.dirname(pbrt_hip.LIB_PATH)


def _build(tmp_path):
    exe = str(tmp_path / "render_box")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "render_box.c"), "-o", exe, "-L" + LIB_DIR, "-lpbrt_hip", "-lm",
           "-Wl,-rpath," + LIB_DIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_compiles_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (covered by the gpu test)")
    r = subprocess.run([exe, str(tmp_path / "out.png")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert "BVH: 11 nodes over 12 triangles" in r.stdout          # the host builder ran
    assert "no CPU fallback" in r.stderr
    assert not (tmp_path / "out.png").exists()


@pytest.mark.gpu
def test_c_caller_renders(tmp_path):
    exe = _build(tmp_path)
    out = tmp_path / "out.png"
    r = subprocess.run([exe, str(out), "96", "64", "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = re.search(r"(\d+) camera samples, (\d+) closest-hit \+ (\d+) shadow rays.*mean RGB ([0-9.]+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == 96 * 64 * 32
    assert int(m.group(2)) > int(m.group(1)) and int(m.group(3)) > 0
    assert 0.05 < float(m.group(4)) < 2.0                          # a lit, closed box
    assert "one rank over RCCL: the merged film equals the one-call film" in r.stdout, r.stdout + r.stderr   # film_create / render_device / film_reduce / film_download from C
    raw = out.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    import struct
    assert struct.unpack(">II", raw[16:24]) == (96, 64)
