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


def _build(tmp_path, source="render_box.cpp"):
    exe = str(tmp_path / source.replace(".cpp", "_cpp"))
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", source), "-o", exe, "-L" + LIB_DIR, "-lpbrt_hip", "-Wl,-rpath," + LIB_DIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_mirror_names_the_reference_surface():
    """Same names as the traits and constructors it stands in for, each with the reference file:line it mirrors."""
    hpp = open(os.path.join(ROOT, "include", "pbrt_hip.hpp")).read()
    for name in ("class Primitive", "class BVHAccel : public Primitive", "class Scene", "class Integrator", "class SamplerIntegrator : public Integrator",
                 "class PathIntegrator : public SamplerIntegrator", "class DirectLightingIntegrator : public SamplerIntegrator", "class Film",
                 "class PerspectiveCamera", "struct RandomSampler", "PbrtInstance TransformedPrimitive(", "class WhittedIntegrator", "class AOIntegrator", "bool intersect(Ray& ray, SurfaceInteraction* isect) const",
                 "bool intersect_p(const Ray& ray) const", "Bounds3f world_bound() const", "void render(const Scene& scene)", "void write_image(",
                 "struct Filter", "struct BoxFilter : Filter", "struct TriangleFilter : Filter", "struct GaussianFilter : Filter", "struct MitchellFilter : Filter",
                 "struct LanczosSincFilter : Filter", "class Camera", "class PerspectiveCamera : public Camera", "class OrthographicCamera : public Camera",
                 "class EnvironmentCamera : public Camera", "struct Sampler", "struct RandomSampler : Sampler", "struct StratifiedSampler : Sampler",
                 "struct ZeroTwoSequenceSampler : Sampler", "struct HaltonSampler : Sampler"):
        assert name in hpp, name
    for cite in ("src/core/primitive.rs:17-30", "src/accelerators/bvh.rs:216-271", "src/core/scene.rs:18-46", "src/core/integrator.rs:29-42",
                 "src/integrators/path.rs:31-46", "src/core/film.rs:30-63", "src/cameras/perspective.rs:34-82", "bvh.rs:934-953",
                 "src/core/filter.rs:10-15", "orthographic.rs:37-80", "environment.rs:19-29", "stratified.rs:22-40", "zerotwosequence.rs:17-23", "halton.rs:63-98"):
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


def test_cpp_variants_caller_compiles_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = _build(tmp_path, "render_variants.cpp")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (covered by the gpu test)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "pbrt::Error (3)" in r.stderr and "no CPU fallback" in r.stderr and r.stdout == ""


def _open_box():
    """The scene of examples/render_variants.cpp (= render_box.c), array for array."""
    import numpy as np
    from pbrt_hip import scenes
    c = {k: (float(2 * int(k[0]) - 1), float(2 * int(k[1]) - 1), float(2 * int(k[2]) - 1)) for k in ("000", "100", "010", "110", "001", "101", "011", "111")}
    quads = [((c["000"], c["100"], c["101"], c["001"]), 0, False), ((c["010"], c["011"], c["111"], c["110"]), 0, False),
             ((c["001"], c["101"], c["111"], c["011"]), 0, False), ((c["000"], c["001"], c["011"], c["010"]), 1, False),
             ((c["100"], c["110"], c["111"], c["101"]), 2, False),
             (((-0.3, 0.99, -0.3), (0.3, 0.99, -0.3), (0.3, 0.99, 0.3), (-0.3, 0.99, 0.3)), 0, True)]
    pos, idx, mat, tri_light, lights = [], [], [], [], []
    for corners, m, emitter in quads:
        v0 = len(pos)
        pos.extend(corners)
        for t in ((0, 1, 2), (0, 2, 3)):
            idx.append([v0 + t[0], v0 + t[1], v0 + t[2]])
            mat.append(m)
            if emitter:
                tri_light.append(len(lights))
                lights.append((scenes.LIGHT_DIFFUSE_AREA, (17.0, 17.0, 17.0), len(idx) - 1, 1, 1))
            else:
                tri_light.append(-1)
    materials = scenes._materials([(scenes.MAT_MATTE, kd, (0, 0, 0), 1.0) for kd in ((0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15))])
    return dict(positions=np.asarray(pos, dtype=np.float32), indices=np.asarray(idx, dtype=np.int32), tri_material=np.asarray(mat, dtype=np.int32),
                materials=materials, tri_light=np.asarray(tri_light, dtype=np.int32), lights=scenes._lights(lights))


@pytest.mark.gpu
def test_cpp_variants_are_the_jobs_the_python_binding_sets_up(tmp_path):
    """Filters, samplers, cameras and integrators named as the reference names them (include/pbrt_hip.hpp) arrive at the C ABI
    as the same jobs the Python binding of that ABI sets up: equal ray counts, equal films (their sums, in double)."""
    import numpy as np
    from pbrt_hip import scenes
    W, H = 64, 48
    r = subprocess.run([_build(tmp_path, "render_variants.cpp"), str(W), str(H)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = {m.group(1): (int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6)))
           for m in re.finditer(r"variant (\S+): (\d+) camera samples, (\d+) closest-hit \+ (\d+) shadow rays; film xyz (\S+) weight (\S+)", r.stdout)}
    cams = {m.group(1): np.array([float.fromhex(v) for v in m.group(2).split()], dtype=np.float32) for m in re.finditer(r"camera (\S+):((?: \S+){32})", r.stdout)}
    assert len(got) == 15 and set(cams) == {"orthographic", "environment"}, r.stdout
    ctx = pbrt_hip.Context(0)
    scene = pbrt_hip.Scene(ctx, _open_box())
    eye, look, up = (0.0, 0.0, -3.4), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    persp = scenes.perspective_camera(eye, look, up, 40.0, W, H)

    def same(name, film, st):
        n, closest, shadow, xyz, weight = got[name]
        assert (st["camera_samples"], st["rays_closest"], st["rays_shadow"]) == (n, closest, shadow), (name, st, got[name])
        f = film.astype(np.float64)
        # wider filters splat with float atomics in whatever order the waves arrive: equal up to the order of the additions
        tol = 1e-6 if name in ("triangle", "gaussian", "mitchell", "lanczos") else 1e-9
        assert abs(f[..., :3].sum() - xyz) <= tol * abs(xyz) and abs(f[..., 3].sum() - weight) <= tol * abs(weight), (name, f[..., :3].sum(), xyz)

    for name, ft in (("box", None), ("triangle", pbrt_hip.filter_table("triangle", 2.0, 1.5)), ("gaussian", pbrt_hip.filter_table("gaussian", 2.0, 2.0, 2.0)),
                     ("mitchell", pbrt_hip.filter_table("mitchell", 2.0, 2.0, 1.0 / 3.0, 1.0 / 3.0)), ("lanczos", pbrt_hip.filter_table("lanczos", 3.0, 3.0, 3.0))):
        same(name, *scene.render(persp, W, H, 16, max_depth=5, seed=7, light_strategy=1, filter=ft))
    for name, smp, spp in (("stratified", ("stratified", 4, 4, True, 4), 16), ("zerotwo", ("zerotwo", 4), 16), ("halton", ("halton",), 16)):
        same(name, *scene.render(persp, W, H, spp, max_depth=4, seed=3, light_strategy=2, sampler=smp))

    def camera_from(values, like):
        cam = like.copy()
        cam["camera_to_world"], cam["raster_to_camera"] = values[:16], values[16:]
        return cam
    ortho = scenes.orthographic_camera(eye, look, up, 1.2, W, H)
    env = scenes.environment_camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), up)
    for key, cam in (("orthographic", ortho), ("environment", env)):   # the C++ constructors against the Python ones, then bit for bit the C++ values
        ref = np.concatenate([np.asarray(cam["camera_to_world"]).reshape(-1), np.asarray(cam["raster_to_camera"]).reshape(-1)])
        np.testing.assert_allclose(cams[key], ref, rtol=0, atol=1e-6)
    same("orthographic-direct", *scene.render(camera_from(cams["orthographic"], ortho), W, H, 8, integrator=pbrt_hip.INTEGRATOR_DIRECT, max_depth=3, seed=1, light_strategy=0))
    same("environment-whitted", *scene.render(camera_from(cams["environment"], env), W, H, 4, integrator=pbrt_hip.INTEGRATOR_WHITTED, max_depth=3, seed=2, light_strategy=0))
    same("environment-ao", *scene.render(camera_from(cams["environment"], env), W, H, 4, integrator=pbrt_hip.INTEGRATOR_AO, ao_samples=16, cos_sample=True, seed=2, max_depth=0))
    # BVHAccel::new(HLBVH) on the device, a Sphere beside the triangles, per-vertex shading normals
    assert "hlbvh device against host: films equal, world bound y [-1.00, 1.00] / [-1.00, 1.00]" in r.stdout, r.stdout
    box = _open_box()
    on_device = pbrt_hip.Scene(ctx, box, device_build=True)
    same("hlbvh-device", *on_device.render(persp, W, H, 16, max_depth=5, seed=5, light_strategy=2))
    on_device.close()
    with_ball = pbrt_hip.Scene(ctx, dict(box, spheres=np.array([[0.2, -0.6, 0.1, 0.4, 1, -1, 0, 0]], dtype=np.float32)))
    same("sphere", *with_ball.render(persp, W, H, 8, integrator=pbrt_hip.INTEGRATOR_DIRECT, max_depth=3, seed=9, light_strategy=1))
    with_ball.close()
    assert re.search(r"sphere: hit 1 t 2\.7000 primitive 12", r.stdout), r.stdout       # z = 0.1 - 0.4 seen from z = -3; 12 triangles come first
    p32 = box["positions"]
    smooth = pbrt_hip.Scene(ctx, dict(box, normals=(-p32 / np.sqrt((p32 * p32).sum(axis=1, dtype=np.float32))[:, None]).astype(np.float32)))
    same("vertex-normals", *smooth.render(persp, W, H, 8, max_depth=3, seed=11, light_strategy=2))
    smooth.close()
    # the general top level: instances of two aggregates, world triangles with the area lights beside them
    inst = np.zeros((3, 2, 4, 4), dtype=np.float32)
    for i, m4 in enumerate(([[1, 0, 0, 3], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], [[0, 0, 1, -3], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]],
                            [[1, 0, 0, 0], [0, 1, 0, -0.5], [0, 0, 1, 0], [0, 0, 0, 1]])):
        inst[i, 0], inst[i, 1] = np.array(m4, dtype=np.float64), np.linalg.inv(np.array(m4, dtype=np.float64))
    two = dict(objects=[dict(positions=box["positions"], indices=box["indices"][:10], tri_material=box["tri_material"][:10]),
                        dict(positions=np.array([[-0.5, 0, -0.5], [0.5, 0, -0.5], [0.5, 0, 0.5], [-0.5, 0, 0.5]], dtype=np.float32),
                             indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32), tri_material=np.array([1, 1], dtype=np.int32))],
               instances=inst, instance_object=np.array([0, 0, 1], dtype=np.int32), instance_material=np.array([-1, 2, -1], dtype=np.int32),
               world=dict(positions=np.array([[-8, -1.25, -8], [8, -1.25, -8], [8, -1.25, 8], [-8, -1.25, 8], [-1, 3, -1], [1, 3, -1], [1, 3, 1], [-1, 3, 1]], dtype=np.float32),
                          indices=np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7]], dtype=np.int32), tri_material=np.zeros(4, dtype=np.int32),
                          tri_light=np.array([-1, -1, 0, 1], dtype=np.int32)),
               materials=box["materials"], lights=scenes._lights([(scenes.LIGHT_DIFFUSE_AREA, (30.0, 30.0, 30.0), t, 0, 1) for t in (2, 3)]))
    top = pbrt_hip.Scene(ctx, two)
    same("two-level", *top.render(scenes.perspective_camera((0.0, 1.5, -9.0), look, up, 40.0, W, H), W, H, 8, max_depth=4, seed=13, light_strategy=2))
    top.close()
    assert re.search(r"two-level: hit 1 t 1\.5000 instance 2 primitive [01]; world bound x \[-8\.0, 8\.0\]", r.stdout), r.stdout
    m = re.search(r"two shares: (\d+) rays against (\d+) of the whole frame, (\d+) of (\d+) film values differ", r.stdout)
    assert m and m.group(1) == m.group(2) and m.group(3) == "0" and int(m.group(4)) == W * H * 4, r.stdout
    assert "world of one over RCCL: film equal" in r.stdout, r.stdout   # pbrt::Comm + the ABI's own device film (pbrt_hip_film_create / _reduce / _download)
    scene.close()
    ctx.close()
