"""Definitions of the small renders behind tests/golden/variants_48x32.npz (one fixture per widened row: sibling
integrators, delta lights, samplers, spatial light sampling, per-vertex shading data, spheres, filters, cameras).
Shared by make_golden.py (which writes the fixture from the oracle) and by the CPU / GPU tests that replay it."""
import numpy as np

from pbrt_hip import scenes

W, H, SPP = 48, 32, 4


def _mixed_with_lights():
    extra = [scenes.point_light((0.2, 0.9, -0.4), (3.0, 3.0, 3.0)), scenes.distant_light((0.0, 1.0, 0.2), (0.8, 0.8, 0.8)),
             scenes.spot_light((1.5, 1.5, 1.5), (0.0, 0.0, 0.0), (20.0, 18.0, 15.0), 40.0, 30.0)]
    return scenes.with_lights(scenes.mixed_materials_scene(), extra)


def _spheres():
    sc = scenes.cornell_box()
    n_tris, n_l = sc["indices"].shape[0], len(sc["lights"])
    sph = np.zeros((2, 8), dtype=np.float32)
    sph[0] = (400.0, 300.0, 200.0, 40.0, 0, n_l, 0, 0)
    sph[1] = (150.0, 80.0, 350.0, 80.0, 1, -1, 0, 0)
    sc["spheres"] = sph
    sc["lights"] = np.concatenate([sc["lights"], scenes._lights([(scenes.LIGHT_DIFFUSE_AREA, (30.0, 25.0, 20.0), n_tris, 0, 1)])])
    return sc


def variants():
    """name -> (scene, camera, render kwargs incl. optional ("filter_spec", kind, radius, a, b))."""
    mixed_cam = scenes.random_triangles_camera(W, H)
    cornell_cam = scenes.cornell_camera(W, H)
    v = {}
    v["whitted_delta_lights"] = (_mixed_with_lights(), mixed_cam, dict(integrator=2, max_depth=4, seed=1))
    v["direct_all_delta_lights"] = (_mixed_with_lights(), mixed_cam, dict(integrator=1, max_depth=3, light_strategy=0, seed=2))
    v["ao_cosine"] = (scenes.cornell_box(), cornell_cam, dict(integrator=3, ao_samples=8, cos_sample=True, seed=3))
    v["path_stratified"] = (_mixed_with_lights(), mixed_cam, dict(max_depth=5, light_strategy=1, seed=4,
                                                                   sampler=("stratified", 2, 2, True, 4)))
    v["path_zerotwo"] = (_mixed_with_lights(), mixed_cam, dict(max_depth=5, light_strategy=1, seed=5, sampler=("zerotwo", 4)))
    v["path_halton"] = (_mixed_with_lights(), mixed_cam, dict(max_depth=5, light_strategy=1, seed=6, sampler=("halton",)))
    v["path_spatial"] = (scenes.with_lights(scenes.cornell_box(), scenes.cornell_delta_lights()), cornell_cam,
                         dict(max_depth=4, light_strategy=2, seed=7))
    v["vertex_shading"] = (scenes.with_vertex_shading(scenes.mixed_materials_scene(), seq=7, tangents=True), mixed_cam,
                           dict(max_depth=5, light_strategy=1, seed=8))
    v["spheres"] = (_spheres(), cornell_cam, dict(max_depth=5, light_strategy=1, seed=9))
    v["gaussian_filter"] = (scenes.cornell_box(), cornell_cam, dict(max_depth=4, seed=10, filter_spec=("gaussian", 1.5, 2.0, 0.0),
                                                                    max_sample_luminance=4.0))
    v["orthographic_lens"] = (scenes.cornell_box(), scenes.orthographic_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0),
                                                                               (0.0, 1.0, 0.0), 280.0, W, H, 15.0, 1100.0),
                              dict(max_depth=4, seed=11))
    v["environment_camera"] = (scenes.cornell_box(), scenes.environment_camera((278.0, 273.0, 200.0), (278.0, 273.0, 500.0),
                                                                               (0.0, 1.0, 0.0)), dict(max_depth=4, seed=12))
    return v


def oracle_scene(oracle, sc):
    return oracle.OracleScene(sc, normals=sc.get("normals"), uvs=sc.get("uvs"), tangents=sc.get("tangents"))
