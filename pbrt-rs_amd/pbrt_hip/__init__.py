"""ctypes binding of libpbrt_hip.so (C ABI: include/pbrt_hip.h).

Host-side mirror of the reference's interface for the hot path, used by tests and bench.py:
  BVHAccel.build       BVHAccel::new             src/accelerators/bvh.rs:216-271 (host)
  Scene                Scene::new                src/core/scene.rs:18-34
  Scene.intersect/_p   Scene::intersect[_p]      src/core/scene.rs:40-46   (batch form)
  Scene.render         Integrator::render        src/core/integrator.rs:399-480
There is no CPU fallback: without the HIP library or without a GPU every compute call raises.
"""
import ctypes
import os
import weakref

import numpy as np

from . import scenes
from .scenes import (CAMERA_DTYPE, HIT_DTYPE, INSTANCE_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE, NODE_DTYPE, RAY_DTYPE)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRT_HIP_LIB") or os.path.join(_HERE, "libpbrt_hip.so")   # PBRT_HIP_LIB: a variant build (development A/Bs)

SPLIT_SAH, SPLIT_HLBVH, SPLIT_MIDDLE, SPLIT_EQUAL_COUNTS = 0, 1, 2, 3
INTEGRATOR_PATH, INTEGRATOR_DIRECT, INTEGRATOR_WHITTED, INTEGRATOR_AO = 0, 1, 2, 3
SAMPLER_RANDOM, SAMPLER_STRATIFIED, SAMPLER_ZEROTWO, SAMPLER_HALTON = 0, 1, 2, 3
TRAVERSAL_AUTO, TRAVERSAL_STACK, TRAVERSAL_STACKLESS = 0, 1, 2   # pbrt_hip_context_set_traversal
WIDE_BUILD_DEVICE, WIDE_BUILD_HOST, WIDE_BUILD_NONE = 0, 1, 2       # pbrt_hip_context_set_wide_build
WIDE_LAYOUT_AUTO, WIDE_LAYOUT_PACKED, WIDE_LAYOUT_LINES = 0, 1, 2    # pbrt_hip_context_set_wide_layout
TILE_ORDER_MORTON, TILE_ORDER_ROW_MAJOR = 0, 1                     # PbrtRenderParams.tile_order

EXPORTS = [
    "pbrt_hip_context_create", "pbrt_hip_context_destroy", "pbrt_hip_last_error", "pbrt_hip_bvh_build",
    "pbrt_hip_free", "pbrt_hip_bvh_build_boxes", "pbrt_hip_instance_bounds", "pbrt_hip_scene_create",
    "pbrt_hip_scene_create_instanced", "pbrt_hip_scene_create_with_spheres", "pbrt_hip_scene_destroy", "pbrt_hip_intersect",
    "pbrt_hip_intersect_p", "pbrt_hip_intersect_device", "pbrt_hip_intersect_p_device", "pbrt_hip_synchronize",
    "pbrt_hip_context_set_deadline", "pbrt_hip_context_set_traversal", "pbrt_hip_context_is_lost", "pbrt_hip_context_set_wide_build", "pbrt_hip_context_set_wide_layout", "pbrt_hip_scene_wide_stride",
    "pbrt_hip_trace_timing", "pbrt_hip_set_counting", "pbrt_hip_get_counters", "pbrt_hip_render", "pbrt_hip_render_device", "pbrt_hip_film_to_rgb",
    "pbrt_hip_bvh_build_hlbvh_device", "pbrt_hip_scene_create_hlbvh", "pbrt_hip_scene_set_shading_data", "pbrt_hip_tile_partition", "pbrt_hip_tile_partition_order", "pbrt_hip_filter_table", "pbrt_hip_sample_bounds", "pbrt_hip_write_pfm", "pbrt_hip_write_png", "pbrt_hip_write_exr",
    "pbrt_hip_comm_unique_id", "pbrt_hip_comm_create", "pbrt_hip_comm_destroy", "pbrt_hip_film_reduce",
    "pbrt_hip_film_create", "pbrt_hip_film_download", "pbrt_hip_film_destroy",
    "pbrt_hip_comm_last_error", "pbrt_hip_scene_wide_records", "pbrt_hip_get_wide_counters", "pbrt_hip_probe_gather", "pbrt_hip_probe_state_stream",
    "pbrt_hip_li", "pbrt_hip_li_device", "pbrt_hip_camera_rays", "pbrt_hip_scene_create_two_level", "pbrt_hip_debug_wide_export",
]


class RenderParams(ctypes.Structure):
    _fields_ = [("integrator", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("rr_threshold", ctypes.c_float),
                ("light_strategy", ctypes.c_int32), ("spp", ctypes.c_int32), ("width", ctypes.c_int32),
                ("height", ctypes.c_int32), ("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32),
                ("y1", ctypes.c_int32), ("seed", ctypes.c_uint64), ("tile_rank", ctypes.c_int32),
                ("tile_world", ctypes.c_int32), ("spp_per_pass", ctypes.c_int32), ("ao_samples", ctypes.c_int32),
                ("filter_radius", ctypes.c_float * 2), ("filter_table", ctypes.c_void_p),
                ("sampler", ctypes.c_int32), ("sampler_x", ctypes.c_int32), ("sampler_y", ctypes.c_int32),
                ("sampler_jitter", ctypes.c_int32), ("sampler_dims", ctypes.c_int32), ("max_sample_luminance", ctypes.c_float),
                ("shade_order", ctypes.c_int32), ("ray_order", ctypes.c_int32), ("tile_order", ctypes.c_int32),
                ("samples_per_wave", ctypes.c_int32)]


class PbrtObject(ctypes.Structure):
    _fields_ = [("positions", ctypes.c_void_p), ("n_verts", ctypes.c_int32), ("indices", ctypes.c_void_p), ("n_tris", ctypes.c_int32),
                ("tri_material", ctypes.c_void_p), ("nodes", ctypes.c_void_p), ("n_nodes", ctypes.c_int32), ("prim_order", ctypes.c_void_p)]


class LiParams(ctypes.Structure):
    _fields_ = [("integrator", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("rr_threshold", ctypes.c_float),
                ("light_strategy", ctypes.c_int32), ("ao_samples", ctypes.c_int32), ("draws_before_li", ctypes.c_int32)]


class RenderStats(ctypes.Structure):
    _fields_ = [("camera_samples", ctypes.c_uint64), ("rays_closest", ctypes.c_uint64),
                ("rays_shadow", ctypes.c_uint64), ("trace_launches", ctypes.c_uint64),
                ("trace_ms", ctypes.c_double), ("total_ms", ctypes.c_double)]


class PbrtHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libpbrt_hip.so. Raises if it has not been built (pbrt-rs_amd/build.sh)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PbrtHipError(f"{LIB_PATH} is missing: build it with pbrt-rs_amd/build.sh "
                               "(__graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        L.pbrt_hip_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.pbrt_hip_context_destroy.argtypes = [vp]
        L.pbrt_hip_context_destroy.restype = None
        L.pbrt_hip_last_error.argtypes = [vp]
        L.pbrt_hip_last_error.restype = ctypes.c_char_p
        L.pbrt_hip_bvh_build.argtypes = [vp, i32, vp, i32, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i32),
                                         ctypes.POINTER(vp)]
        L.pbrt_hip_bvh_build_hlbvh_device.argtypes = [vp, vp, i32, vp, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i32),
                                                      ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_double)]
        L.pbrt_hip_scene_create_hlbvh.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, i32, i32, ctypes.POINTER(vp),
                                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.pbrt_hip_scene_set_shading_data.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
        L.pbrt_hip_scene_create_with_spheres.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, vp, i32, vp,
                                                         ctypes.POINTER(vp)]
        L.pbrt_hip_free.argtypes = [vp]
        L.pbrt_hip_free.restype = None
        L.pbrt_hip_scene_create.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp,
                                            ctypes.POINTER(vp)]
        L.pbrt_hip_bvh_build_boxes.argtypes = [vp, vp, i32, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i32),
                                               ctypes.POINTER(vp)]
        L.pbrt_hip_instance_bounds.argtypes = [vp, vp, vp, i32, vp, vp]
        L.pbrt_hip_scene_create_instanced.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp, vp, i32,
                                                      vp, i32, vp, ctypes.POINTER(vp)]
        L.pbrt_hip_scene_create_two_level.argtypes = [vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, i32, vp,
                                                      ctypes.POINTER(vp)]
        L.pbrt_hip_scene_destroy.argtypes = [vp]
        L.pbrt_hip_scene_destroy.restype = None
        L.pbrt_hip_scene_wide_records.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_char_p)]
        for name in ("pbrt_hip_intersect", "pbrt_hip_intersect_p", "pbrt_hip_intersect_device",
                     "pbrt_hip_intersect_p_device"):
            getattr(L, name).argtypes = [vp, vp, i64, vp]
        L.pbrt_hip_comm_unique_id.argtypes = [vp]
        L.pbrt_hip_comm_create.argtypes = [vp, i32, i32, vp, ctypes.POINTER(vp)]
        L.pbrt_hip_comm_destroy.argtypes = [vp]
        L.pbrt_hip_comm_destroy.restype = None
        L.pbrt_hip_film_reduce.argtypes = [vp, vp, i64, i32]
        L.pbrt_hip_film_create.argtypes = [vp, i64, ctypes.POINTER(vp)]
        L.pbrt_hip_film_download.argtypes = [vp, vp, i64, vp]
        L.pbrt_hip_film_destroy.argtypes = [vp, vp]
        L.pbrt_hip_film_destroy.restype = None
        L.pbrt_hip_comm_last_error.argtypes = []
        L.pbrt_hip_comm_last_error.restype = ctypes.c_char_p
        L.pbrt_hip_synchronize.argtypes = [vp]
        L.pbrt_hip_context_set_deadline.argtypes = [vp, ctypes.c_double]
        L.pbrt_hip_context_is_lost.argtypes = [vp]
        L.pbrt_hip_trace_timing.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_uint64)]
        L.pbrt_hip_set_counting.argtypes = [vp, ctypes.c_int]
        L.pbrt_hip_context_set_traversal.argtypes = [vp, ctypes.c_int]
        L.pbrt_hip_context_set_wide_build.argtypes = [vp, ctypes.c_int]
        L.pbrt_hip_context_set_wide_layout.argtypes = [vp, ctypes.c_int]
        L.pbrt_hip_scene_wide_stride.argtypes = [vp]
        L.pbrt_hip_get_counters.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64 * 4)]
        L.pbrt_hip_get_wide_counters.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64 * 4)]
        L.pbrt_hip_probe_gather.argtypes = [vp, i64, i32, i32, i32, ctypes.POINTER(ctypes.c_double)]
        L.pbrt_hip_probe_state_stream.argtypes = [vp, i64, i32, i64, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]
        L.pbrt_hip_render.argtypes = [vp, vp, ctypes.POINTER(RenderParams), vp, ctypes.POINTER(RenderStats)]
        L.pbrt_hip_render_device.argtypes = [vp, vp, ctypes.POINTER(RenderParams), vp, ctypes.POINTER(RenderStats)]
        L.pbrt_hip_li.argtypes = [vp, ctypes.POINTER(LiParams), vp, vp, i64, vp, ctypes.POINTER(RenderStats)]
        L.pbrt_hip_li_device.argtypes = [vp, ctypes.POINTER(LiParams), vp, vp, i64, vp, ctypes.POINTER(RenderStats)]
        L.pbrt_hip_camera_rays.argtypes = [vp, vp, ctypes.POINTER(RenderParams), i64, vp, vp, vp, vp, ctypes.POINTER(i64)]
        L.pbrt_hip_tile_partition.argtypes = [i32, i32, i32, i32, i32, i32, vp, i32, ctypes.POINTER(i32)]
        L.pbrt_hip_tile_partition_order.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, i32, ctypes.POINTER(i32)]
        L.pbrt_hip_filter_table.argtypes = [i32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
        L.pbrt_hip_sample_bounds.argtypes = [i32, i32, ctypes.c_float, ctypes.c_float, vp]
        L.pbrt_hip_write_pfm.argtypes = [ctypes.c_char_p, vp, i32, i32]
        L.pbrt_hip_write_png.argtypes = [ctypes.c_char_p, vp, i32, i32]
        L.pbrt_hip_write_exr.argtypes = [ctypes.c_char_p, vp, i32, i32]
        L.pbrt_hip_film_to_rgb.argtypes = [vp, i64, vp]
        L.pbrt_hip_film_to_rgb.restype = None
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """One GPU (one process per GPU)."""

    def __init__(self, device_id=0):
        L = lib()
        h = ctypes.c_void_p()
        rc = L.pbrt_hip_context_create(device_id, ctypes.byref(h))
        if rc != 0:
            raise PbrtHipError(f"pbrt_hip_context_create failed ({rc}): {L.pbrt_hip_last_error(None).decode()}")
        self.h = h
        self._scenes = weakref.WeakSet()  # scenes are destroyed before their context

    def check(self, rc, what):
        if rc != 0:
            raise PbrtHipError(f"{what} failed ({rc}): {lib().pbrt_hip_last_error(self.h).decode()}")

    def synchronize(self):
        self.check(lib().pbrt_hip_synchronize(self.h), "synchronize")

    def set_deadline(self, seconds):
        """Seconds pbrt_hip_render / pbrt_hip_li wait for one wavefront before the context is given up for lost."""
        self.check(lib().pbrt_hip_context_set_deadline(self.h, float(seconds)), "set_deadline")

    def is_lost(self):
        """True once a call on this context ran into the wavefront deadline: its device memory only goes back with the process."""
        return lib().pbrt_hip_context_is_lost(self.h) == 1

    def trace_timing(self, reset=False):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self.check(lib().pbrt_hip_trace_timing(self.h, int(reset), ctypes.byref(ms), ctypes.byref(n)), "trace_timing")
        return ms.value, n.value

    def set_traversal(self, traversal):
        """TRAVERSAL_AUTO (4-wide records where the scene has them), TRAVERSAL_STACK (binary records, per-lane stack) or
        TRAVERSAL_STACKLESS (binary records, parent links + bit trail; single-level triangle scenes only)."""
        self.check(lib().pbrt_hip_context_set_traversal(self.h, int(traversal)), "context_set_traversal")

    def set_wide_build(self, where):
        """Where scenes created from now on get their 4-wide records: WIDE_BUILD_DEVICE (default), WIDE_BUILD_HOST (the host
        builder: same bytes) or WIDE_BUILD_NONE (binary records only)."""
        self.check(lib().pbrt_hip_context_set_wide_build(self.h, int(where)), "context_set_wide_build")

    def set_wide_layout(self, layout):
        """How scenes created from now on keep their 4-wide records and triangles: WIDE_LAYOUT_AUTO (packed while they fit 8 MiB,
        else one 64-byte line each), WIDE_LAYOUT_PACKED, WIDE_LAYOUT_LINES."""
        self.check(lib().pbrt_hip_context_set_wide_layout(self.h, int(layout)), "context_set_wide_layout")

    def set_counting(self, enable):
        """Instrumented traversal. True / 1: box / triangle test counts of the reference's loops (binary kernels);
        2: the wide kernels' own record / leaf / triangle fetch counts (wide_counters). Slow."""
        self.check(lib().pbrt_hip_set_counting(self.h, int(enable)), "set_counting")

    def wide_counters(self, reset=False):
        c = (ctypes.c_uint64 * 4)()
        self.check(lib().pbrt_hip_get_wide_counters(self.h, int(reset), ctypes.byref(c)), "get_wide_counters")
        return dict(records=int(c[0]), leaf_candidates=int(c[1]), triangles=int(c[2]), special_rays=int(c[3]))

    def probe_state_stream(self, n_paths, density=0.7, gather_table_bytes=48 << 20, parts=31):
        """k_shade's access pattern with a known byte count (pbrt_hip_probe_state_stream): (bytes read, bytes written, ms) per launch."""
        rd, wr, ms = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        self.check(lib().pbrt_hip_probe_state_stream(self.h, int(n_paths), int(round(density * 1000)), int(gather_table_bytes), int(parts),
                                                     ctypes.byref(rd), ctypes.byref(wr), ctypes.byref(ms)), "probe_state_stream")
        return rd.value, wr.value, ms.value

    def probe_gather(self, table_bytes, record_bytes=48, waves_per_simd=5, iters=256):
        """Measured rate (records / s) of dependent random record fetches from a table of table_bytes."""
        r = ctypes.c_double()
        self.check(lib().pbrt_hip_probe_gather(self.h, int(table_bytes), int(record_bytes), int(waves_per_simd), int(iters),
                                               ctypes.byref(r)), "probe_gather")
        return r.value

    def counters(self, reset=False):
        c = (ctypes.c_uint64 * 4)()
        self.check(lib().pbrt_hip_get_counters(self.h, int(reset), ctypes.byref(c)), "get_counters")
        d = dict(rays=int(c[0]), node_tests=int(c[1]), prim_tests=int(c[2]))
        if c[3]:
            d["inst_tests"] = int(c[3])
        return d

    def close(self):
        if self.h:
            for sc in list(self._scenes):
                sc.close()
            lib().pbrt_hip_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bvh_build(positions, indices, max_prims_in_node=4, split_method=SPLIT_SAH):
    """BVHAccel::new on the host. Returns (nodes[NODE_DTYPE], prim_order[int32]). Needs no GPU."""
    L = lib()
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    nodes_p, order_p, n_nodes = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32()
    rc = L.pbrt_hip_bvh_build(_p(positions), positions.shape[0], _p(indices), indices.shape[0], max_prims_in_node,
                              split_method, ctypes.byref(nodes_p), ctypes.byref(n_nodes), ctypes.byref(order_p))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_bvh_build failed ({rc})")
    n = n_nodes.value
    if n == 0:
        return np.zeros(0, dtype=NODE_DTYPE), np.zeros(0, dtype=np.int32)
    return _take_tree(nodes_p, n, order_p, indices.shape[0])


def bvh_build_hlbvh_device(ctx, positions, indices, max_prims_in_node=4):
    """BVHAccel::new (HLBVH) on the GPU. Returns (nodes, prim_order, build_ms); equals bvh_build(..., SPLIT_HLBVH)."""
    positions = np.ascontiguousarray(positions, dtype=np.float32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    nodes_p, order_p, n_nodes, ms = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32(), ctypes.c_double()
    rc = lib().pbrt_hip_bvh_build_hlbvh_device(ctx.h, _p(positions), positions.shape[0], _p(indices), indices.shape[0],
                                               max_prims_in_node, ctypes.byref(nodes_p), ctypes.byref(n_nodes),
                                               ctypes.byref(order_p), ctypes.byref(ms))
    ctx.check(rc, "pbrt_hip_bvh_build_hlbvh_device")
    if n_nodes.value == 0:
        return np.zeros(0, dtype=NODE_DTYPE), np.zeros(0, dtype=np.int32), 0.0
    nodes, order = _take_tree(nodes_p, n_nodes.value, order_p, indices.shape[0])
    return nodes, order, ms.value


def _take_tree(nodes_p, n_nodes, order_p, n_prims):
    L = lib()
    try:
        nodes = np.frombuffer(ctypes.string_at(nodes_p, n_nodes * NODE_DTYPE.itemsize), dtype=NODE_DTYPE).copy()
        order = np.frombuffer(ctypes.string_at(order_p, n_prims * 4), dtype=np.int32).copy()
    finally:
        L.pbrt_hip_free(nodes_p)
        L.pbrt_hip_free(order_p)
    return nodes, order


def bvh_build_boxes(bounds_min, bounds_max, max_prims_in_node=4, split_method=SPLIT_SAH):
    """BVHAccel::new over caller-supplied primitive bounds (e.g. instance world bounds). Host only."""
    lo = np.ascontiguousarray(bounds_min, dtype=np.float32)
    hi = np.ascontiguousarray(bounds_max, dtype=np.float32)
    nodes_p, order_p, n_nodes = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32()
    rc = lib().pbrt_hip_bvh_build_boxes(_p(lo), _p(hi), lo.shape[0], max_prims_in_node, split_method,
                                        ctypes.byref(nodes_p), ctypes.byref(n_nodes), ctypes.byref(order_p))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_bvh_build_boxes failed ({rc})")
    if n_nodes.value == 0:
        return np.zeros(0, dtype=NODE_DTYPE), np.zeros(0, dtype=np.int32)
    return _take_tree(nodes_p, n_nodes.value, order_p, lo.shape[0])


def make_instances(scene):
    """scene["instances"] (n,2,4,4) + scene["instance_material"] -> PbrtInstance array."""
    m = np.ascontiguousarray(scene["instances"], dtype=np.float32)
    inst = np.zeros(m.shape[0], dtype=INSTANCE_DTYPE)
    inst["to_world"] = m[:, 0].reshape(-1, 16)
    inst["to_object"] = m[:, 1].reshape(-1, 16)
    inst["material"] = scene["instance_material"]
    return inst


def instance_bounds(object_min, object_max, instances):
    """TransformedPrimitive::world_bound per instance. Host only."""
    n = len(instances)
    lo, hi = np.zeros((n, 3), dtype=np.float32), np.zeros((n, 3), dtype=np.float32)
    omin, omax = np.ascontiguousarray(object_min, dtype=np.float32), np.ascontiguousarray(object_max, dtype=np.float32)
    rc = lib().pbrt_hip_instance_bounds(_p(omin), _p(omax), _p(instances), n, _p(lo), _p(hi))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_instance_bounds failed ({rc})")
    return lo, hi


def build_general_two_level(scene, max_prims_in_node=4, split_method=SPLIT_SAH):
    """Host BVH builds for scenes.two_level_scene-style scenes: one tree per object, the top-level tree over the instances'
    world bounds followed by the world triangles' bounds. Returns (object trees [(nodes, order)], instances, tlas_nodes, tlas_order)."""
    trees = [bvh_build(o["positions"], o["indices"], max_prims_in_node, split_method) for o in scene["objects"]]
    inst = make_instances(scene)
    io = np.asarray(scene["instance_object"], dtype=np.int32)
    lo, hi = np.zeros((len(inst), 3), dtype=np.float32), np.zeros((len(inst), 3), dtype=np.float32)
    for k, (nodes, _) in enumerate(trees):
        sel = np.flatnonzero(io == k)
        if len(sel):
            a, b = instance_bounds(nodes[0]["bmin"], nodes[0]["bmax"], np.ascontiguousarray(inst[sel]))
            lo[sel], hi[sel] = a, b
    w = scene["world"]
    if len(w["indices"]):
        tri = np.asarray(w["positions"], dtype=np.float32)[np.asarray(w["indices"], dtype=np.int32)]
        lo, hi = np.concatenate([lo, tri.min(axis=1)]), np.concatenate([hi, tri.max(axis=1)])
    tlas_nodes, tlas_order = bvh_build_boxes(lo, hi, max_prims_in_node, split_method)
    return trees, inst, tlas_nodes, tlas_order


def build_two_level(scene, max_prims_in_node=4, split_method=SPLIT_SAH, tlas_max_prims=None):
    """Host BVH builds for an instanced scene: object-level tree over the triangles, top-level tree over
    the instances' world bounds. Returns (blas_nodes, blas_order, instances, tlas_nodes, tlas_order)."""
    blas_nodes, blas_order = bvh_build(scene["positions"], scene["indices"], max_prims_in_node, split_method)
    inst = make_instances(scene)
    lo, hi = instance_bounds(blas_nodes[0]["bmin"], blas_nodes[0]["bmax"], inst)
    tlas_nodes, tlas_order = bvh_build_boxes(lo, hi, tlas_max_prims or max_prims_in_node, split_method)
    return blas_nodes, blas_order, inst, tlas_nodes, tlas_order


class Scene:
    """Scene::new: triangles as GeometricPrimitives in a BVHAccel, resident in HBM.
    With scene["instances"]: TransformedPrimitive instances of the triangle aggregate (two levels)."""

    def __init__(self, ctx, scene, max_prims_in_node=4, split_method=SPLIT_SAH, bvh=None, device_build=False):
        """device_build=True: BVHAccel::new(HLBVH) built and laid out on the GPU (pbrt_hip_scene_create_hlbvh);
        self.build_ms / self.layout_ms then hold the HIP-event times and self.nodes is None."""
        self.ctx = ctx
        if "objects" in scene:
            self._init_two_level(scene, max_prims_in_node, split_method, bvh)
            return
        if "instances" in scene:
            self._init_instanced(scene, max_prims_in_node, split_method, bvh)
            return
        if device_build:
            self._init_device_build(scene, max_prims_in_node)
            return
        self.positions = np.ascontiguousarray(scene["positions"], dtype=np.float32)
        self.indices = np.ascontiguousarray(scene["indices"], dtype=np.int32)
        tri_material = np.ascontiguousarray(scene["tri_material"], dtype=np.int32)
        materials = np.ascontiguousarray(scene["materials"], dtype=MATERIAL_DTYPE)
        tri_light = np.ascontiguousarray(scene["tri_light"], dtype=np.int32)
        lights = np.ascontiguousarray(scene["lights"], dtype=LIGHT_DTYPE)
        spheres = scene.get("spheres")
        if spheres is not None and len(spheres):
            self._init_with_spheres(scene, tri_material, materials, tri_light, lights, max_prims_in_node, split_method, bvh)
            return
        if bvh is None:
            bvh = bvh_build(self.positions, self.indices, max_prims_in_node, split_method)
        self.nodes, self.prim_order = bvh
        h = ctypes.c_void_p()
        rc = lib().pbrt_hip_scene_create(ctx.h, _p(self.positions), self.positions.shape[0], _p(self.indices),
                                         self.indices.shape[0], _p(tri_material), _p(materials), len(materials),
                                         _p(tri_light), _p(lights) if len(lights) else None, len(lights),
                                         _p(self.nodes), len(self.nodes), _p(self.prim_order), ctypes.byref(h))
        ctx.check(rc, "pbrt_hip_scene_create")
        self.h = h
        ctx._scenes.add(self)
        self._set_shading_data(scene)

    def _init_with_spheres(self, scene, tri_material, materials, tri_light, lights, max_prims_in_node, split_method, bvh):
        """scene["spheres"]: (n, 8) {centre.xyz, radius, material, light index or -1, 0, 0}; sphere i is primitive n_tris + i."""
        sph = np.ascontiguousarray(scene["spheres"], dtype=np.float32).reshape(-1, 8)
        if bvh is None:
            tri = self.positions[self.indices]                       # Triangle::world_bound (triangle.rs:175-180)
            c, r = sph[:, :3], sph[:, 3:4]
            lo = np.concatenate([tri.min(axis=1), c + (-r)])         # Sphere::world_bound through translate(c)
            hi = np.concatenate([tri.max(axis=1), c + r])
            bvh = bvh_build_boxes(lo, hi, max_prims_in_node, split_method)
        self.nodes, self.prim_order = bvh
        sph4 = np.ascontiguousarray(sph[:, :4])
        sph_mat = np.ascontiguousarray(sph[:, 4].astype(np.int32))
        sph_light = np.ascontiguousarray(sph[:, 5].astype(np.int32))
        h = ctypes.c_void_p()
        rc = lib().pbrt_hip_scene_create_with_spheres(
            self.ctx.h, _p(self.positions), self.positions.shape[0], _p(self.indices), self.indices.shape[0], _p(tri_material),
            _p(materials), len(materials), _p(tri_light), _p(lights) if len(lights) else None, len(lights), _p(sph4),
            _p(sph_mat), _p(sph_light), len(sph4), _p(self.nodes), len(self.nodes), _p(self.prim_order), ctypes.byref(h))
        self.ctx.check(rc, "pbrt_hip_scene_create_with_spheres")
        self.h = h
        self.ctx._scenes.add(self)

    def debug_wide_export(self, n_slots):
        """(records (n, 12) uint32, wide-order triangles (n_slots, 12) float32, leaf boxes (n_slots, 8) float32) of a
        single-level scene, copied back from the device (pbrt_hip_debug_wide_export: the header's diagnostic entry point)."""
        n, _ = self.wide_records()
        nodes = np.zeros((max(n, 0), 12), dtype=np.uint32)
        tris = np.zeros((n_slots, 12), dtype=np.float32)
        boxes = np.zeros((n_slots, 8), dtype=np.float32)
        f = lib().pbrt_hip_debug_wide_export
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        f.restype = ctypes.c_int
        self.ctx.check(f(self.h, nodes.ctypes.data, tris.ctypes.data, boxes.ctypes.data, int(n_slots)), "debug_wide_export")
        return nodes, tris, boxes

    def wide_records(self):
        """(number of 4-wide records, reason): -1 records = the scene is traced over the binary records only."""
        n, why = ctypes.c_int32(), ctypes.c_char_p()
        self.ctx.check(lib().pbrt_hip_scene_wide_records(self.h, ctypes.byref(n), ctypes.byref(why)), "scene_wide_records")
        return n.value, (why.value or b"").decode()

    def wide_stride(self):
        """Bytes between two 4-wide records (and two wide-order triangles) in HBM: 48 packed, 64 one line each, 0 = no wide records."""
        return lib().pbrt_hip_scene_wide_stride(self.h)

    def _set_shading_data(self, scene):
        """TriangleMesh n / uv (triangle.rs:17-26): scene["normals"] / scene["tangents"] (n_verts, 3), scene["uvs"] (n_verts, 2), optional."""
        normals, tangents, uvs = scene.get("normals"), scene.get("tangents"), scene.get("uvs")
        if normals is None and tangents is None and uvs is None:
            return
        normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        tangents = None if tangents is None else np.ascontiguousarray(tangents, dtype=np.float32)
        uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32)
        rc = lib().pbrt_hip_scene_set_shading_data(self.h, _p(self.positions), self.positions.shape[0], _p(self.indices),
                                                   self.indices.shape[0], _p(normals), _p(tangents), _p(uvs))
        self.ctx.check(rc, "pbrt_hip_scene_set_shading_data")

    def _init_device_build(self, scene, max_prims_in_node):
        self.positions = np.ascontiguousarray(scene["positions"], dtype=np.float32)
        self.indices = np.ascontiguousarray(scene["indices"], dtype=np.int32)
        tri_material = np.ascontiguousarray(scene["tri_material"], dtype=np.int32)
        materials = np.ascontiguousarray(scene["materials"], dtype=MATERIAL_DTYPE)
        tri_light = np.ascontiguousarray(scene["tri_light"], dtype=np.int32)
        lights = np.ascontiguousarray(scene["lights"], dtype=LIGHT_DTYPE)
        self.nodes = self.prim_order = None
        h, b_ms, l_ms = ctypes.c_void_p(), ctypes.c_double(), ctypes.c_double()
        rc = lib().pbrt_hip_scene_create_hlbvh(self.ctx.h, _p(self.positions), self.positions.shape[0], _p(self.indices),
                                               self.indices.shape[0], _p(tri_material), _p(materials), len(materials),
                                               _p(tri_light), _p(lights) if len(lights) else None, len(lights),
                                               max_prims_in_node, ctypes.byref(h), ctypes.byref(b_ms), ctypes.byref(l_ms))
        self.ctx.check(rc, "pbrt_hip_scene_create_hlbvh")
        self.h = h
        self.build_ms, self.layout_ms = b_ms.value, l_ms.value
        self.ctx._scenes.add(self)
        self._set_shading_data(scene)

    def _init_instanced(self, scene, max_prims_in_node, split_method, bvh):
        self.positions = np.ascontiguousarray(scene["positions"], dtype=np.float32)
        self.indices = np.ascontiguousarray(scene["indices"], dtype=np.int32)
        tri_material = np.ascontiguousarray(scene["tri_material"], dtype=np.int32)
        materials = np.ascontiguousarray(scene["materials"], dtype=MATERIAL_DTYPE)
        lights = np.ascontiguousarray(scene["lights"], dtype=LIGHT_DTYPE)
        if bvh is None:
            bvh = build_two_level(scene, max_prims_in_node, split_method)
        self.nodes, self.prim_order, self.instances, self.tlas_nodes, self.tlas_order = bvh
        h = ctypes.c_void_p()
        rc = lib().pbrt_hip_scene_create_instanced(
            self.ctx.h, _p(self.positions), self.positions.shape[0], _p(self.indices), self.indices.shape[0],
            _p(tri_material), _p(materials), len(materials), _p(lights) if len(lights) else None, len(lights),
            _p(self.nodes), len(self.nodes), _p(self.prim_order), _p(self.instances), len(self.instances),
            _p(self.tlas_nodes), len(self.tlas_nodes), _p(self.tlas_order), ctypes.byref(h))
        self.ctx.check(rc, "pbrt_hip_scene_create_instanced")
        self.h = h
        self.ctx._scenes.add(self)
        self._set_shading_data(scene)

    def _init_two_level(self, scene, max_prims_in_node, split_method, bvh):
        """scenes.two_level_scene: several object aggregates, instances of them, world-space triangles (area lights) beside."""
        if bvh is None:
            bvh = build_general_two_level(scene, max_prims_in_node, split_method)
        self.object_trees, self.instances, self.tlas_nodes, self.tlas_order = bvh
        keep = []
        objs = (PbrtObject * len(scene["objects"]))()
        for k, (o, (nodes, order)) in enumerate(zip(scene["objects"], self.object_trees)):
            pos = np.ascontiguousarray(o["positions"], dtype=np.float32)
            idx = np.ascontiguousarray(o["indices"], dtype=np.int32)
            mat = np.ascontiguousarray(o["tri_material"], dtype=np.int32)
            keep += [pos, idx, mat, nodes, order]
            objs[k] = PbrtObject(pos.ctypes.data, len(pos), idx.ctypes.data, len(idx), mat.ctypes.data, nodes.ctypes.data, len(nodes),
                                 order.ctypes.data)
        io = np.ascontiguousarray(scene["instance_object"], dtype=np.int32)
        w = scene["world"]
        wp = np.ascontiguousarray(w["positions"], dtype=np.float32)
        wi = np.ascontiguousarray(w["indices"], dtype=np.int32)
        wm = np.ascontiguousarray(w["tri_material"], dtype=np.int32)
        wl = np.ascontiguousarray(w["tri_light"], dtype=np.int32)
        materials = np.ascontiguousarray(scene["materials"], dtype=MATERIAL_DTYPE)
        lights = np.ascontiguousarray(scene["lights"], dtype=LIGHT_DTYPE)
        h = ctypes.c_void_p()
        rc = lib().pbrt_hip_scene_create_two_level(
            self.ctx.h, ctypes.addressof(objs), len(objs), _p(self.instances), _p(io), len(self.instances),
            _p(wp) if len(wi) else None, len(wp) if len(wi) else 0, _p(wi) if len(wi) else None, len(wi),
            _p(wm) if len(wi) else None, _p(wl) if len(wi) else None, _p(materials), len(materials),
            _p(lights) if len(lights) else None, len(lights), _p(self.tlas_nodes), len(self.tlas_nodes), _p(self.tlas_order),
            ctypes.byref(h))
        self.ctx.check(rc, "pbrt_hip_scene_create_two_level")
        self.h = h
        self.ctx._scenes.add(self)

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        self.ctx.check(lib().pbrt_hip_intersect(self.h, _p(rays), len(rays), _p(out)), "pbrt_hip_intersect")
        return out

    def intersect_p(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), dtype=np.uint8)
        self.ctx.check(lib().pbrt_hip_intersect_p(self.h, _p(rays), len(rays), _p(out)), "pbrt_hip_intersect_p")
        return out

    def intersect_device(self, d_rays_ptr, n, d_out_ptr):
        self.ctx.check(lib().pbrt_hip_intersect_device(self.h, ctypes.c_void_p(d_rays_ptr), n,
                                                       ctypes.c_void_p(d_out_ptr)), "pbrt_hip_intersect_device")

    def intersect_p_device(self, d_rays_ptr, n, d_out_ptr):
        self.ctx.check(lib().pbrt_hip_intersect_p_device(self.h, ctypes.c_void_p(d_rays_ptr), n,
                                                         ctypes.c_void_p(d_out_ptr)), "pbrt_hip_intersect_p_device")

    def _params(self, width, height, spp, integrator, max_depth, rr_threshold, light_strategy, seed, bounds,
                tile_rank, tile_world, spp_per_pass, filter=None, ao_samples=64, sampler=None, max_sample_luminance=0.0,
                shade_order=0, ray_order=0, tile_order=0, samples_per_wave=0):
        rx, ry, table = (0.5, 0.5, None) if filter is None else filter
        if table is not None:
            table = np.ascontiguousarray(table, dtype=np.float32)
            self._filter_keep = table
        x0, y0, x1, y1 = bounds if bounds is not None else sample_bounds(width, height, rx, ry)
        if sampler is None:
            smp = (SAMPLER_RANDOM, 0, 0, 0, 0)
        elif sampler[0] == "stratified":       # ("stratified", nx, ny, jitter, n_dims): StratifiedSampler::new
            smp = (SAMPLER_STRATIFIED, sampler[1], sampler[2], int(bool(sampler[3])), sampler[4])
        elif sampler[0] == "zerotwo":          # ("zerotwo", n_dims): ZeroTwoSequenceSampler::new
            smp = (SAMPLER_ZEROTWO, 1, 1, 1, sampler[1])
        elif sampler[0] == "halton":           # ("halton",): HaltonSampler::new over the film's sample bounds
            smp = (SAMPLER_HALTON, 1, 1, 1, 0)
        else:
            raise ValueError(sampler)
        return RenderParams(integrator, max_depth, rr_threshold, light_strategy, spp, width, height, x0, y0, x1, y1,
                            seed, tile_rank, tile_world, spp_per_pass, ao_samples, (ctypes.c_float * 2)(rx, ry),
                            None if table is None else table.ctypes.data, *smp, float(max_sample_luminance), int(shade_order), int(ray_order),
                            int(tile_order), int(samples_per_wave))

    def render(self, camera, width, height, spp, integrator=INTEGRATOR_PATH, max_depth=5, rr_threshold=1.0,
               light_strategy=1, seed=0, bounds=None, tile_rank=0, tile_world=1, spp_per_pass=0, d_film_ptr=None,
               filter=None, ao_samples=64, cos_sample=True, sampler=None, max_sample_luminance=0.0, shade_order=0, ray_order=0,
               tile_order=0, samples_per_wave=0):
        """Integrator::render. Returns (film[h,w,4] or None when d_film_ptr is given, stats dict).
        integrator: INTEGRATOR_PATH / _DIRECT / _WHITTED / _AO (ao_samples, cos_sample: AOIntegrator::new).
        sampler: None (RandomSampler), ("stratified", nx, ny, jitter, n_dims) or ("zerotwo", n_dims); the samples
        per pixel then become nx * ny / the next power of two of spp.
        filter = (radius_x, radius_y, table256) from filter_table(), None = 0.5 box.
        shade_order: 0 queue order, 1 by material inside blocks, 2 sorted queue (PbrtRenderParams.shade_order).
        ray_order: 0 ray queues in Morton order from the second bounce on, 1 queue order (PbrtRenderParams.ray_order).
        tile_order: TILE_ORDER_MORTON (0) / TILE_ORDER_ROW_MAJOR (1): how the 16x16 tiles are dealt to the tile_world ranks.
        samples_per_wave: consecutive samples of a pixel that share a wave (0 = library default: one pixel per wave, 1 = rounds 1-4's layout)."""
        camera = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        if integrator == INTEGRATOR_AO:
            light_strategy = int(bool(cos_sample))
        rp = self._params(width, height, spp, integrator, max_depth, rr_threshold, light_strategy, seed, bounds,
                          tile_rank, tile_world, spp_per_pass, filter, ao_samples, sampler, max_sample_luminance, shade_order, ray_order,
                          tile_order, samples_per_wave)
        st = RenderStats()
        if d_film_ptr is None:
            film = np.zeros((height, width, 4), dtype=np.float32)
            rc = lib().pbrt_hip_render(self.h, _p(camera), ctypes.byref(rp), _p(film), ctypes.byref(st))
        else:
            film = None
            rc = lib().pbrt_hip_render_device(self.h, _p(camera), ctypes.byref(rp), ctypes.c_void_p(d_film_ptr),
                                              ctypes.byref(st))
        self.ctx.check(rc, "pbrt_hip_render")
        stats = {name: getattr(st, name) for name, _ in RenderStats._fields_}
        return film, stats

    def li(self, rays, stream_keys, integrator=INTEGRATOR_PATH, max_depth=5, rr_threshold=1.0, light_strategy=1, ao_samples=64,
           draws_before_li=0):
        """Integrator::li for a batch of rays (RAY_DTYPE) with one RandomSampler stream each (uint64 keys).
        Returns (rgb[n, 3], stats)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        keys = np.ascontiguousarray(stream_keys, dtype=np.uint64)
        assert len(rays) == len(keys)
        rgb = np.zeros((len(rays), 3), dtype=np.float32)
        lp = LiParams(integrator, max_depth, rr_threshold, light_strategy, ao_samples, draws_before_li)
        st = RenderStats()
        self.ctx.check(lib().pbrt_hip_li(self.h, ctypes.byref(lp), _p(rays), _p(keys), len(rays), _p(rgb), ctypes.byref(st)), "pbrt_hip_li")
        return rgb, {name: getattr(st, name) for name, _ in RenderStats._fields_}

    def camera_rays(self, camera, width, height, spp, seed=0, bounds=None, tile_rank=0, tile_world=1, tile_order=0, sampler=None, filter=None):
        """The camera-ray stage of render(): (rays[RAY_DTYPE], stream_keys[uint64], p_film[n, 2], pixel_sample[n, 3]).
        sampler as in render(): the camera sample is the sampler's first 2D, first 1D (ray.time) and second 2D draw (sampler.rs:66-73)."""
        camera = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        rp = self._params(width, height, spp, INTEGRATOR_PATH, 5, 1.0, 1, seed, bounds, tile_rank, tile_world, 0, filter, 64, sampler, 0.0,
                          tile_order=tile_order)
        n = ctypes.c_int64()
        lib().pbrt_hip_camera_rays(self.h, _p(camera), ctypes.byref(rp), 0, None, None, None, None, ctypes.byref(n))
        rays = np.zeros(n.value, dtype=RAY_DTYPE)
        keys = np.zeros(n.value, dtype=np.uint64)
        pfilm = np.zeros((n.value, 2), dtype=np.float32)
        pix = np.zeros((n.value, 3), dtype=np.int32)
        self.ctx.check(lib().pbrt_hip_camera_rays(self.h, _p(camera), ctypes.byref(rp), n.value, _p(rays), _p(keys), _p(pfilm), _p(pix),
                                                  ctypes.byref(n)), "pbrt_hip_camera_rays")
        return rays, keys, pfilm, pix

    def close(self):
        if getattr(self, "h", None):
            if self.ctx.h:
                lib().pbrt_hip_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId: 128 bytes that rank 0 hands to the other ranks out of band."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    rc = lib().pbrt_hip_comm_unique_id(buf)
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_comm_unique_id failed ({rc}): {lib().pbrt_hip_comm_last_error().decode()}")
    return bytes(buf)


class Comm:
    """RCCL communicator of one rank (one process per GPU) for the film merge."""

    def __init__(self, ctx, world, rank, unique_id):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self.ctx, self.world, self.rank = ctx, world, rank
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        rc = lib().pbrt_hip_comm_create(ctx.h, world, rank, buf, ctypes.byref(h))
        if rc != 0:
            raise PbrtHipError(f"pbrt_hip_comm_create failed ({rc}): {lib().pbrt_hip_comm_last_error().decode()}")
        self.h = h

    def film_reduce(self, d_film_ptr, n_pixels, root=0):
        """Sum of the ranks' device films in place; root < 0: all-reduce."""
        rc = lib().pbrt_hip_film_reduce(self.h, ctypes.c_void_p(d_film_ptr), n_pixels, root)
        if rc != 0:
            raise PbrtHipError(f"pbrt_hip_film_reduce failed ({rc}): {lib().pbrt_hip_comm_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            lib().pbrt_hip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceFilm:
    """A film on the context's device (pbrt_hip_film_create): what Scene.render(d_film_ptr=film.ptr) renders into and
    Comm.film_reduce merges, for a host without a device allocator of its own. download() -> [height, width, 4] float32."""

    def __init__(self, ctx, width, height):
        self.ctx, self.width, self.height = ctx, width, height
        h = ctypes.c_void_p()
        ctx.check(lib().pbrt_hip_film_create(ctx.h, width * height, ctypes.byref(h)), "pbrt_hip_film_create")
        self.ptr = h.value
        ctx._scenes.add(self)   # closed with the scenes, before the context

    def download(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self.ctx.check(lib().pbrt_hip_film_download(self.ctx.h, ctypes.c_void_p(self.ptr), self.width * self.height, _p(out)), "pbrt_hip_film_download")
        return out

    def close(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            lib().pbrt_hip_film_destroy(self.ctx.h, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_partition(bounds, rank, world, order=0):
    """16x16 tile origins of `bounds` = (x0, y0, x1, y1) owned by `rank`: the k-th tile of the dealing order (TILE_ORDER_MORTON,
    the default, or TILE_ORDER_ROW_MAJOR) belongs to rank k % world; returned in row-major order. Host only."""
    x0, y0, x1, y1 = bounds
    n = ctypes.c_int32()
    rc = lib().pbrt_hip_tile_partition_order(x0, y0, x1, y1, rank, world, order, None, 0, ctypes.byref(n))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_tile_partition_order failed ({rc})")
    out = np.zeros((n.value, 2), dtype=np.int32)
    rc = lib().pbrt_hip_tile_partition_order(x0, y0, x1, y1, rank, world, order, _p(out), n.value, ctypes.byref(n))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_tile_partition_order failed ({rc})")
    return out


FILTERS = dict(box=0, gaussian=1, mitchell=2, lanczos=3, triangle=4)


def filter_table(kind, rx, ry, a=0.0, b=0.0):
    """(rx, ry, table256) for Scene.render(filter=...): Film::new's table of the named reconstruction filter."""
    t = np.zeros(256, dtype=np.float32)
    rc = lib().pbrt_hip_filter_table(FILTERS[kind], rx, ry, a, b, _p(t))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_filter_table failed ({rc})")
    return (float(rx), float(ry), t)


def sample_bounds(width, height, rx=0.5, ry=0.5):
    b = np.zeros(4, dtype=np.int32)
    rc = lib().pbrt_hip_sample_bounds(width, height, rx, ry, _p(b))
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_sample_bounds failed ({rc})")
    return tuple(int(v) for v in b)


def write_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    rc = lib().pbrt_hip_write_pfm(str(path).encode(), _p(rgb), rgb.shape[1], rgb.shape[0])
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_write_pfm failed ({rc})")


def write_exr(path, rgb):
    """Uncompressed float32 OpenEXR of an RGB float image (h, w, 3), linear values. Host only."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    rc = lib().pbrt_hip_write_exr(str(path).encode(), _p(rgb), rgb.shape[1], rgb.shape[0])
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_write_exr failed ({rc})")


def write_png(path, rgb):
    """8-bit sRGB PNG of an RGB float image (h, w, 3). Host only."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    rc = lib().pbrt_hip_write_png(str(path).encode(), _p(rgb), rgb.shape[1], rgb.shape[0])
    if rc != 0:
        raise PbrtHipError(f"pbrt_hip_write_png failed ({rc})")


def film_to_rgb(film):
    film = np.ascontiguousarray(film, dtype=np.float32)
    rgb = np.zeros(film.shape[:-1] + (3,), dtype=np.float32)
    lib().pbrt_hip_film_to_rgb(_p(film), film.size // 4, _p(rgb))
    return rgb
