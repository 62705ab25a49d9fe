"""include/pbrt_hip.hpp — the host side above the C ABI in C++, mirroring the reference's trait surface for the path
(Primitive / BVHAccel / Scene / Integrator / PathIntegrator / Film / PerspectiveCamera): examples/render_box.cpp
compiles as strict C++17 against it, links to the in-tree library and (on a GPU box) renders the very film the C99 caller of
the same scene renders, answers Primitive::intersect / intersect_p ray by ray with the closed-form values, and Integrator::li
for one ray; without a GPU it stops at Context creation with pbrt::Error (no CPU fallback)."""
import os
import re
import subprocess

import pytest

import pbrt_hip
from test_c_example import _build as build_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.dirname(pbrt_hip.LIB_PATH)


def _build(tmp_path):
    exe = str(tmp_path / "render_box_cpp")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "render_box.cpp"), "-o", exe, "-L" + LIB_DIR, "-lpbrt_hip", "-Wl,-rpath," + LIB_DIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_mirror_names_the_reference_surface():
    """Same names as the traits and constructors it stands in for, each with the reference file:line it mirrors."""
    hpp = open(os.path.join(ROOT, "include", "pbrt_hip.hpp")).read()
    for name in ("class Primitive", "class BVHAccel : public Primitive", "class Scene", "class Integrator", "class SamplerIntegrator : public Integrator",
                 "class PathIntegrator : public SamplerIntegrator", "class DirectLightingIntegrator : public SamplerIntegrator", "class Film",
                 "class PerspectiveCamera", "struct RandomSampler", "PbrtInstance TransformedPrimitive(", "class WhittedIntegrator", "class AOIntegrator", "bool intersect(Ray& ray, SurfaceInteraction* isect) const",
                 "bool intersect_p(const Ray& ray) const", "Bounds3f world_bound() const", "void render(const Scene& scene)", "void write_image("):
        assert name in hpp, name
    for cite in ("src/core/primitive.rs:17-30", "src/accelerators/bvh.rs:216-271", "src/core/scene.rs:18-46", "src/core/integrator.rs:29-42",
                 "src/integrators/path.rs:31-46", "src/core/film.rs:30-63", "src/cameras/perspective.rs:34-82", "bvh.rs:934-953"):
        assert cite in hpp, cite
    assert "#include <torch" not in hpp and "hip_runtime" not in hpp and "#include <hip" not in hpp   # plain C++ over the C ABI


def test_cpp_caller_compiles_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (covered by the gpu test)")
    r = subprocess.run([exe, str(tmp_path / "out.png")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "pbrt::Error (3)" in r.stderr and "no CPU fallback" in r.stderr
    assert not (tmp_path / "out.png").exists()


@pytest.mark.gpu
def test_cpp_caller_renders_what_the_c_caller_renders(tmp_path):
    exe = _build(tmp_path)
    out = tmp_path / "out.png"
    r = subprocess.run([exe, str(out), "96", "64", "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = re.search(r"(\d+) camera samples, (\d+) closest-hit \+ (\d+) shadow rays.*mean RGB ([0-9.]+)", r.stdout)
    assert m and int(m.group(1)) == 96 * 64 * 32, r.stdout
    assert "BVH: 11 nodes over 12 triangles" in r.stdout
    # the C99 caller of the same scene, camera, integrator, sampler seed: the same rays, the same film
    c = subprocess.run([build_c(tmp_path), str(tmp_path / "c.png"), "96", "64", "32"], capture_output=True, text=True, timeout=300)
    mc = re.search(r"(\d+) camera samples, (\d+) closest-hit \+ (\d+) shadow rays.*mean RGB ([0-9.]+)", c.stdout)
    assert mc and m.group(1, 2, 3, 4) == mc.group(1, 2, 3, 4), (r.stdout, c.stdout)
    assert out.read_bytes() == (tmp_path / "c.png").read_bytes()
    # Primitive::intersect from the centre of the box straight down: the floor y = -1 at t = 1 exactly, ray.t_max lowered to it,
    # one of the floor's two triangles, barycentrics of the centre of the quad's diagonal (b0 = b2 = 0.5 or b1 = 0 ...)
    hit = re.search(r"intersect: hit (\d) t ([0-9.]+) ray.t_max ([0-9.]+) primitive (-?\d+) barycentrics (-?[0-9.]+) (-?[0-9.]+) (-?[0-9.]+)", r.stdout)
    assert hit and hit.group(1) == "1" and float(hit.group(2)) == 1.0 and float(hit.group(3)) == 1.0 and int(hit.group(4)) in (0, 1)
    assert abs(sum(float(hit.group(k)) for k in (5, 6, 7)) - 1.0) < 1e-6
    p = re.search(r"intersect_p: blocked-short (\d) towards-floor (\d) through-the-opening (\d)", r.stdout)
    # a shadow ray stopping short of the floor, the same ray once its t_max IS the hit distance (Bounds3f::intersect_p wants
    # t_min < ray.t_max, geometry.rs:748: the floor at exactly t_max is not in front of it), a ray through the open side
    assert p and p.groups() == ("0", "0", "0")
    li = re.search(r"li towards the emitter: ([0-9.]+) ([0-9.]+) ([0-9.]+)", r.stdout)
    assert li and all(float(v) >= 17.0 for v in li.groups())      # Le of the emitter it looks at, plus what the path gathers after it
    assert "aggregate.get_material(): Primitive::get_material: aggregates do not hold one" in r.stdout
    # TransformedPrimitives under a top-level BVHAccel (primitive.rs:105-159): the floor of the copy moved to x = +5, seen from its
    # centre, is at t = 1 inside instance 0; the world bound spans both copies; nothing between them
    assert "instanced: hit 1 t 1.000000 instance 0 world bound x [-6.0, 6.0] miss-between 0" in r.stdout, r.stdout
