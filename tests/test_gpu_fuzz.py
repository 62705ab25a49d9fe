"""A slice of the fuzz campaigns of tools/fuzz_intersect.py and tools/fuzz_render.py in the suite (the full campaigns:
profiles/r03_fuzz.txt — 225 500 / 626 500 cases). Random shared-vertex meshes with rays aimed at their vertices and edges, wide,
binary and stackless kernels against the oracle bit for bit; random small scenes through every integrator / sampler / light kind, films against
the oracle. Seeds are fixed, so a failure names its case."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tool,n_cases,first", [("fuzz_intersect.py", 400, 100), ("fuzz_render.py", 1500, 50_000)])
def test_fuzz_slice(tool, n_cases, first):
    # a child process: the campaigns own their context and exit non-zero on the first report
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(n_cases), str(first)], cwd=ROOT, capture_output=True,
                       text=True, timeout=580)
    tail = "\n".join(ln for ln in r.stdout.splitlines() if not ln.startswith("..."))[-3000:]
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert f"{n_cases} cases" in tail and "0 mismatching" in tail
