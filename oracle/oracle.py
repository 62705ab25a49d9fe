"""ctypes loader for the CPU oracle (oracle/liboracle.so).

ORACLE — TEST INFRASTRUCTURE ONLY. May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, and nowhere else; the product path (pbrt-rs_amd/) never touches it.
Parity unpinned: the reference (lazytiger/pbrt-rs) holds no golden vectors for this path and
can be neither built nor run here (no rustc/cargo), so this restatement is
pinned only by its own known-answer and analytic tests (tests/test_oracle_*.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

QUIRKS = dict(D2=1 << 0, D9=1 << 1, D10=1 << 2, D11=1 << 3, D13=1 << 4, D36=1 << 5, D37=1 << 6, D39=1 << 7)


def build(force=False):
    """make is incremental: a library older than its sources is rebuilt (a stale checker would check nothing)."""
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    except (OSError, subprocess.CalledProcessError):
        if force or not os.path.exists(_LIB_PATH):
            raise
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp, ip, vp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
        L.orc_scene_create_with_spheres.restype = vp
        L.orc_scene_set_tangents.argtypes = [vp, vp, ctypes.c_int]
        L.orc_scene_set_tangents.restype = None
        L.orc_scene_create_with_spheres.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp,
                                                    vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_uint32]
        L.orc_scene_create_instanced.restype = vp
        L.orc_scene_create_instanced.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp,
                                                 ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_uint32]
        L.orc_scene_create_two_level.restype = vp
        L.orc_scene_create_two_level.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp,
                                                 ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_uint32]
        L.orc_scene_num_object_nodes.argtypes = [vp, ctypes.c_int]
        L.orc_scene_get_object.argtypes = [vp, ctypes.c_int, vp, vp]
        L.orc_scene_num_blas_nodes.argtypes = [vp]
        L.orc_scene_get_blas.argtypes = [vp, vp, vp]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_num_nodes.argtypes = [vp]
        L.orc_scene_get_nodes.argtypes = [vp, vp]
        L.orc_scene_get_prim_order.argtypes = [vp, vp]
        L.orc_intersect.argtypes = [vp, vp, ctypes.c_int64, vp, vp, ctypes.c_int]
        L.orc_intersect_p.argtypes = [vp, vp, ctypes.c_int64, vp, vp, ctypes.c_int]
        L.orc_render.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_uint64] + [ctypes.c_int] * 7 + [vp, vp]
        L.orc_render_filtered.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_uint64] + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        L.orc_li.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, vp, vp, ctypes.c_int64, ctypes.c_int, vp, vp]
        L.orc_filter_table.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
        L.orc_sample_bounds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp]
        L.orc_triangle_test.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp]
        L.orc_bounds_intersect_p.argtypes = [vp, vp, ctypes.c_uint32]
        L.orc_bounds_intersect_p.restype = ctypes.c_int
        L.orc_offset_ray_origin.argtypes = [vp, vp, vp, vp, vp]
        for name in ("orc_next_float_up", "orc_next_float_down", "orc_gamma"):
            getattr(L, name).restype = ctypes.c_float
            getattr(L, name).argtypes = [ctypes.c_float]
        L.orc_pcg32.argtypes = [ctypes.c_uint64, ctypes.c_int, vp, vp]
        L.orc_elementary.argtypes = [ctypes.c_int, vp, vp, ctypes.c_int64, vp]
        L.orc_sample.argtypes = [ctypes.c_int, vp, ctypes.c_int64, ctypes.c_uint32, vp]
        L.orc_fr_dielectric.restype = ctypes.c_float
        L.orc_fr_dielectric.argtypes = [ctypes.c_float] * 3
        L.orc_node_visits.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, vp]
        L.orc_node_visits.restype = None
        L.orc_sphere_test.argtypes = [vp, ctypes.c_float, vp, vp]
        L.orc_sphere_test.restype = None
        L.orc_refract.argtypes = [vp, vp, ctypes.c_float, ctypes.c_uint32, vp]
        L.orc_refract.restype = ctypes.c_int
        L.orc_local_to_world.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _materials_flat(materials):
    m = np.zeros((len(materials), 8), dtype=np.float32)
    m[:, 0] = materials["type"]
    m[:, 1:4] = materials["kd"]
    m[:, 4:7] = materials["kt"]
    m[:, 7] = materials["eta"]
    return m


def _lights_flat(lights):
    """24 floats per light: type, L rgb, prim, two_sided, n_samples, 0, pos xyz, cos_total_width,
    cos_falloff_start, world_to_light 3x3, 0, 0 (the field order of PbrtLight)."""
    l = np.zeros((len(lights), 24), dtype=np.float32)
    l[:, 0] = lights["type"]
    l[:, 1:4] = lights["L"]
    l[:, 4] = lights["prim"]
    l[:, 5] = lights["two_sided"]
    l[:, 6] = lights["n_samples"]
    if lights.dtype.names and "pos" in lights.dtype.names:
        l[:, 8:11] = lights["pos"]
        l[:, 11] = lights["cos_total_width"]
        l[:, 12] = lights["cos_falloff_start"]
        l[:, 13:22] = lights["world_to_light"]
    return l


HIT_DTYPE = np.dtype([("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("prim_id", "<i4"),
                      ("instance_id", "<i4"), ("pad", "<i4", 2)])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("n_primitives", "<u2"),
                       ("axis", "u1"), ("pad", "u1")])


class OracleScene:
    """Scene::new over GeometricPrimitive triangles in a BVHAccel (CPU restatement)."""

    def __init__(self, scene, max_prims_in_node=4, split_method=0, quirks=0, normals=None, uvs=None, tangents=None):
        L = lib()
        if "objects" in scene:
            self._init_two_level(scene, max_prims_in_node, split_method, quirks)
            return
        if "instances" in scene:
            self._init_instanced(scene, max_prims_in_node, split_method, quirks)
            return
        self._keep = dict(
            positions=_f32(scene["positions"]), indices=np.ascontiguousarray(scene["indices"], dtype=np.int32),
            tri_material=np.ascontiguousarray(scene["tri_material"], dtype=np.int32),
            materials=_materials_flat(scene["materials"]),
            tri_light=np.ascontiguousarray(scene["tri_light"], dtype=np.int32),
            lights=_lights_flat(scene["lights"]),
            normals=None if normals is None else _f32(normals), uvs=None if uvs is None else _f32(uvs))
        k = self._keep
        self.n_tris = k["indices"].shape[0]
        spheres = scene.get("spheres")
        k["spheres"] = None if spheres is None else _f32(spheres).reshape(-1, 8)
        n_spheres = 0 if spheres is None else k["spheres"].shape[0]
        self.n_prims = self.n_tris + n_spheres
        self.h = L.orc_scene_create_with_spheres(
            _p(k["positions"]), k["positions"].shape[0], _p(k["indices"]), self.n_tris, _p(k["normals"]), _p(k["uvs"]),
            _p(k["tri_material"]), _p(k["materials"]), len(k["materials"]), _p(k["tri_light"]), _p(k["lights"]),
            len(k["lights"]), _p(k["spheres"]), n_spheres, max_prims_in_node, split_method, quirks)
        if tangents is not None:
            k["tangents"] = _f32(tangents)
            L.orc_scene_set_tangents(self.h, _p(k["tangents"]), k["positions"].shape[0])

    def _init_two_level(self, scene, max_prims_in_node, split_method, quirks):
        """scenes.two_level_scene: several object aggregates, instances of them, world-space triangles beside (area lights)."""
        from pbrt_hip import scenes as _scenes
        pos, idx, mat, lgt, off = _scenes.combined_two_level_mesh(scene)
        L = lib()
        self._keep = dict(
            positions=pos, indices=idx, tri_material=mat, tri_light=lgt, off=off,
            instances=_f32(scene["instances"]).reshape(-1, 32),
            instance_object=np.ascontiguousarray(scene["instance_object"], dtype=np.int32),
            instance_material=np.ascontiguousarray(scene["instance_material"], dtype=np.int32),
            materials=_materials_flat(scene["materials"]), lights=_lights_flat(scene["lights"]))
        k = self._keep
        self.n_tris = idx.shape[0]
        self.n_instances = k["instances"].shape[0]
        self.obj_tri_offset = off
        self.n_world_tris = len(scene["world"]["indices"])
        self.h = L.orc_scene_create_two_level(_p(pos), pos.shape[0], _p(idx), self.n_tris, _p(off), len(off) - 1, self.n_world_tris,
                                              _p(mat), _p(lgt), _p(k["instances"]), _p(k["instance_object"]),
                                              _p(k["instance_material"]), self.n_instances, _p(k["materials"]), len(k["materials"]),
                                              _p(k["lights"]), len(k["lights"]), max_prims_in_node, split_method, quirks)

    def object_tree(self, k):
        n = lib().orc_scene_num_object_nodes(self.h, k)
        nodes = np.zeros(n, dtype=NODE_DTYPE)
        order = np.zeros(int(self.obj_tri_offset[k + 1] - self.obj_tri_offset[k]), dtype=np.int32)
        lib().orc_scene_get_object(self.h, k, _p(nodes), _p(order))
        return nodes, order   # BVHAccel's leaf order indexes the primitive list it was built over: the object's own triangles

    def _local_prim_ids(self, hits):
        """combined-mesh triangle ids -> index inside the hit instance's object / among the world triangles (PbrtHit's meaning)."""
        if not hasattr(self, "obj_tri_offset"):
            return hits
        prim, inst = hits["prim_id"], hits["instance_id"]
        obj = self._keep["instance_object"][np.maximum(inst, 0)]
        base = np.where(inst >= 0, self.obj_tri_offset[obj], self.obj_tri_offset[-1])
        hits["prim_id"] = np.where(prim >= 0, prim - base, -1)
        return hits

    def _init_instanced(self, scene, max_prims_in_node, split_method, quirks):
        """Config 5: TransformedPrimitive instances of one base mesh (scene["instances"] = (n,2,4,4) float32
        {to_world, to_object}, scene["instance_material"] = (n,) int32)."""
        L = lib()
        self._keep = dict(
            positions=_f32(scene["positions"]), indices=np.ascontiguousarray(scene["indices"], dtype=np.int32),
            instances=_f32(scene["instances"]).reshape(-1, 32),
            instance_material=np.ascontiguousarray(scene["instance_material"], dtype=np.int32),
            tri_material=np.ascontiguousarray(scene.get("tri_material", np.zeros(len(scene["indices"]), dtype=np.int32)), dtype=np.int32),
            materials=_materials_flat(scene["materials"]), lights=_lights_flat(scene["lights"]))
        k = self._keep
        self.n_tris = k["indices"].shape[0]
        self.n_instances = k["instances"].shape[0]
        self.h = L.orc_scene_create_instanced(_p(k["positions"]), k["positions"].shape[0], _p(k["indices"]), self.n_tris, _p(k["tri_material"]),
                                              _p(k["instances"]), _p(k["instance_material"]), self.n_instances,
                                              _p(k["materials"]), len(k["materials"]), _p(k["lights"]),
                                              len(k["lights"]), max_prims_in_node, split_method, quirks)

    def blas(self):
        n = lib().orc_scene_num_blas_nodes(self.h)
        nodes = np.zeros(n, dtype=NODE_DTYPE)
        order = np.zeros(self.n_tris, dtype=np.int32)
        lib().orc_scene_get_blas(self.h, _p(nodes), _p(order))
        return nodes, order

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nodes(self):
        n = lib().orc_scene_num_nodes(self.h)
        out = np.zeros(n, dtype=NODE_DTYPE)
        lib().orc_scene_get_nodes(self.h, _p(out))
        return out

    def prim_order(self):
        n = getattr(self, "n_instances", getattr(self, "n_prims", self.n_tris)) + getattr(self, "n_world_tris", 0)
        out = np.zeros(n, dtype=np.int32)
        lib().orc_scene_get_prim_order(self.h, _p(out))
        return out

    def intersect(self, rays, n_threads=8):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        ctr = np.zeros(4, dtype=np.uint64)
        lib().orc_intersect(self.h, _p(rays), len(rays), _p(out), _p(ctr), n_threads)
        return self._local_prim_ids(out), self._ctr(ctr)

    def intersect_p(self, rays, n_threads=8):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), dtype=np.uint8)
        ctr = np.zeros(4, dtype=np.uint64)
        lib().orc_intersect_p(self.h, _p(rays), len(rays), _p(out), _p(ctr), n_threads)
        return out, self._ctr(ctr)

    def _ctr(self, ctr):
        d = dict(rays=int(ctr[0]), node_tests=int(ctr[1]), prim_tests=int(ctr[2]))
        if hasattr(self, "n_instances"):
            d["inst_tests"] = int(ctr[3])
        return d

    def render(self, cam36, width, height, spp, integrator=0, max_depth=5, rr_threshold=1.0, light_strategy=1,
               seed=0, bounds=None, n_threads=8, filter=None, ao_samples=64, cos_sample=True, sampler=None,
               max_sample_luminance=0.0):
        """integrator: 0 path, 1 direct lighting, 2 Whitted, 3 ambient occlusion (ao_samples, cos_sample).
        sampler: None (random) or ("stratified", nx, ny, jitter, n_dims) / ("zerotwo", n_dims) / ("halton",); spp then becomes
        nx * ny / the next power of two.
        filter = (radius_x, radius_y, table[256]) or None for the 0.5 box; with a wider filter the default
        bounds are Film::get_sample_bounds (pixels outside the film are sampled too)."""
        film = np.zeros((height, width, 4), dtype=np.float32)
        stats = np.zeros(6, dtype=np.uint64)
        cam36 = _f32(cam36)
        rx, ry, table = (0.5, 0.5, None) if filter is None else filter
        table = None if table is None else _f32(table)
        x0, y0, x1, y1 = bounds if bounds is not None else sample_bounds(width, height, rx, ry)
        if integrator == 3:
            max_depth, light_strategy = ao_samples, int(bool(cos_sample))
        lib().orc_render_filtered(self.h, _p(cam36), integrator, max_depth, rr_threshold, light_strategy, spp, seed,
                                  width, height, x0, y0, x1, y1, n_threads, rx, ry, _p(table), _p(sampler_spec(sampler, max_sample_luminance)),
                                  _p(film), _p(stats))
        st = dict(rays=int(stats[0]), node_tests=int(stats[1]), prim_tests=int(stats[2]),
                  camera_samples=int(stats[3]), seconds=float(stats[4]) * 1e-9)
        if hasattr(self, "n_instances"):
            st["inst_tests"] = int(stats[5])
        return film, st


def _li(self, rays, stream_keys, integrator=0, max_depth=5, rr_threshold=1.0, light_strategy=1, ao_samples=64, cos_sample=True,
        draws_before_li=0):
    """Integrator::li for caller-supplied rays (structured o, d, t_max, time records) and RandomSampler stream keys."""
    rays = np.ascontiguousarray(rays)
    assert rays.dtype.itemsize == 32
    keys = np.ascontiguousarray(stream_keys, dtype=np.uint64)
    rgb = np.zeros((len(rays), 3), dtype=np.float32)
    stats = np.zeros(3, dtype=np.uint64)
    if integrator == 3:
        max_depth, light_strategy = ao_samples, int(bool(cos_sample))
    lib().orc_li(self.h, integrator, max_depth, rr_threshold, light_strategy, _p(rays), _p(keys), len(rays),
                 draws_before_li, _p(rgb), _p(stats))
    return rgb, dict(rays=int(stats[0]), node_tests=int(stats[1]), prim_tests=int(stats[2]))


OracleScene.li = _li


def sampler_spec(sampler, max_sample_luminance=0.0):
    """("stratified", nx, ny, jitter, n_dims) / ("zerotwo", n_dims) / ("halton",) / None -> int32[6]; the last entry
    carries the float bits of Film::max_sample_luminance (0 = infinity)."""
    if sampler is None:
        spec = [0, 1, 1, 1, 0]
    elif sampler[0] == "stratified":
        spec = [1, sampler[1], sampler[2], int(bool(sampler[3])), sampler[4]]
    elif sampler[0] == "zerotwo":
        spec = [2, 1, 1, 1, sampler[1]]
    elif sampler[0] == "halton":
        spec = [3, 1, 1, 1, 0]
    else:
        raise ValueError(sampler)
    out = np.zeros(6, dtype=np.int32)
    out[:5] = spec
    out[5:6] = np.array([max_sample_luminance], dtype=np.float32).view(np.int32)
    return out


def brute_force(positions, indices, rays, to_object=None, n_threads=8):
    """Closest hit of every ray over ALL triangles (of every instance given by its world-to-object matrix), no aggregate:
    (t[n] (inf = miss), prim[n], instance[n], ties[n] = triangles reaching exactly the smallest t)."""
    positions, indices = _f32(positions), np.ascontiguousarray(indices, dtype=np.int32)
    r = np.ascontiguousarray(rays)   # RAY_DTYPE: o[3], d[3], t_max, time = 8 floats
    assert r.dtype.itemsize == 32
    n = len(r)
    t, prim, inst, ties = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    m = None if to_object is None else np.ascontiguousarray(to_object, dtype=np.float32).reshape(-1, 16)
    L = lib()
    L.orc_brute_force.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_brute_force.restype = None
    L.orc_brute_force(_p(positions), _p(indices), len(indices), _p(m), 0 if m is None else len(m), _p(r), n, _p(t), _p(prim), _p(inst), _p(ties), n_threads)
    return t, prim, inst, ties


def sampler_tables(sampler, spp, seed=0, pixel_index=0):
    """The PixelSampler tables of one pixel after start_pixel: (samples_1d[n_dims, spp'], samples_2d[n_dims, spp', 2]) with
    spp' = the samples per pixel the sampler takes. Stratified and (0,2) samplers only."""
    spec = sampler_spec(sampler)
    n_dims = int(spec[4])
    cap = max(int(spec[1]) * int(spec[2]), 1)
    while cap < spp:
        cap *= 2
    cap = max(cap, spp) * 2
    a, b = np.zeros((n_dims, cap), dtype=np.float32), np.zeros((n_dims, cap, 2), dtype=np.float32)
    L = lib()
    L.orc_sampler_tables.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = L.orc_sampler_tables(_p(spec), spp, seed, pixel_index, None, None, 0)
    if n <= 0:
        raise ValueError("no tables for this sampler")
    a, b = np.zeros((n_dims, n), dtype=np.float32), np.zeros((n_dims, n, 2), dtype=np.float32)
    assert L.orc_sampler_tables(_p(spec), spp, seed, pixel_index, _p(a), _p(b), n) == n
    return a, b


FILTERS = dict(box=0, gaussian=1, mitchell=2, lanczos=3, triangle=4)


def filter_table(kind, rx, ry, a=0.0, b=0.0):
    """Film::new's 16x16 table of filter.evaluate (film.rs:52-63). a, b: gaussian alpha / mitchell B, C / lanczos tau."""
    t = np.zeros(256, dtype=np.float32)
    lib().orc_filter_table(FILTERS[kind], rx, ry, a, b, _p(t))
    return t


def sample_bounds(width, height, rx=0.5, ry=0.5):
    out = np.zeros(4, dtype=np.int32)
    lib().orc_sample_bounds(width, height, rx, ry, _p(out))
    return tuple(int(v) for v in out)


def film_to_rgb(film):
    """Film::write_image arithmetic (src/core/film.rs:153-178): max(0, xyz_to_rgb(xyz) / weight)."""
    xyz = film[..., :3].astype(np.float32)
    w = film[..., 3:4]
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556],
                  [0.055648, -0.204043, 1.057311]], dtype=np.float32)
    rgb = np.stack([m[i, 0] * xyz[..., 0] + m[i, 1] * xyz[..., 1] + m[i, 2] * xyz[..., 2] for i in range(3)], axis=-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(w != 0, np.maximum(rgb * (np.float32(1.0) / w), 0), rgb)
    return out.astype(np.float32)
