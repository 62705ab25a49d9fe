"""pbrt_hip_intersect_device / pbrt_hip_intersect_p_device: the batch calls over the CALLER's device buffers (what a host with a
device allocator of its own uses to stay off PCIe). Same kernels as the host-buffer calls: the same hits bit for
bit, on flat, instanced and two-level scenes; asynchronous until pbrt_hip_synchronize; empty batches and null pointers."""
import ctypes

import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


class DeviceBuffer:
    """A buffer from the HIP runtime libpbrt_hip.so itself runs on (no second ROCm stack in the process)."""

    def __init__(self, host=None, nbytes=0):
        self.hip = ctypes.CDLL("libamdhip64.so.7")
        self.hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self.hip.hipFree.argtypes = [ctypes.c_void_p]
        self.nbytes = host.nbytes if host is not None else nbytes
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), max(self.nbytes, 16)) == 0
        self.ptr = p.value
        if host is not None:
            assert self.hip.hipMemcpy(self.ptr, host.ctypes.data, self.nbytes, 1) == 0   # host to device
        else:
            assert self.hip.hipMemset(self.ptr, 0xA5, max(self.nbytes, 16)) == 0

    def numpy(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert self.hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0         # device to host, synchronous
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def _scene_cases():
    flat = scenes.random_triangles(20000, seq=3, extent=1.0, size=0.06)
    return {"flat": (flat, scenes.random_rays(50000, 5, origin_extent=1.5)),
            "instanced": (scenes.instanced_scene(500, 20, extent=1.0, base_extent=0.4, tri_size=0.1), scenes.random_rays(30000, 6, origin_extent=1.5)),
            "two-level": (scenes.two_level_scene(30), scenes.random_rays(30000, 7, origin_extent=3.0))}


@pytest.mark.parametrize("name", ["flat", "instanced", "two-level"])
def test_device_buffer_calls_equal_the_host_buffer_calls(hip_ctx, name):
    sc, rays = _scene_cases()[name]
    rays = np.ascontiguousarray(rays, dtype=pbrt_hip.RAY_DTYPE)
    rays["t_max"][::3] = 1.2                                        # a third of them segments
    g = pbrt_hip.Scene(hip_ctx, sc)
    want, want_p = g.intersect(rays), g.intersect_p(rays)
    assert (want["prim_id"] >= 0).sum() > len(rays) // 20
    d_rays = DeviceBuffer(rays)
    d_hits, d_any = DeviceBuffer(nbytes=32 * len(rays)), DeviceBuffer(nbytes=len(rays))
    g.intersect_device(d_rays.ptr, len(rays), d_hits.ptr)
    g.intersect_p_device(d_rays.ptr, len(rays), d_any.ptr)
    hip_ctx.synchronize()                                           # the device calls only queue the work
    got = d_hits.numpy(pbrt_hip.HIT_DTYPE, len(rays))
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(d_any.numpy(np.uint8, len(rays)), want_p)
    assert np.array_equal(d_rays.numpy(pbrt_hip.RAY_DTYPE, len(rays)).view(np.uint8), rays.view(np.uint8))   # the rays are not written
    # a prefix of the batch; an empty batch touches nothing
    d_part = DeviceBuffer(nbytes=32 * 1000)
    g.intersect_device(d_rays.ptr, 1000, d_part.ptr)
    hip_ctx.synchronize()
    assert d_part.numpy(pbrt_hip.HIT_DTYPE, 1000).tobytes() == want[:1000].tobytes()
    g.intersect_device(d_rays.ptr, 0, d_part.ptr)
    g.intersect_p_device(0, 0, 0)
    hip_ctx.synchronize()
    assert d_part.numpy(pbrt_hip.HIT_DTYPE, 1000).tobytes() == want[:1000].tobytes()
    for call, args in ((g.intersect_device, (0, 10, d_part.ptr)), (g.intersect_device, (d_rays.ptr, 10, 0)), (g.intersect_p_device, (d_rays.ptr, -1, d_any.ptr))):
        with pytest.raises(pbrt_hip.PbrtHipError):
            call(*args)
    for b in (d_rays, d_hits, d_any, d_part):
        b.free()
    g.close()


def test_trace_timing_accumulates_what_the_render_calls_report(hip_ctx):
    """pbrt_hip_trace_timing: the context's running total of HIP-event time in the traversal kernel and of its launches = the sum
    of what every render call reported (PbrtRenderStats.trace_ms / trace_launches); reading with reset starts over."""
    g = pbrt_hip.Scene(hip_ctx, scenes.cornell_box())
    cam = scenes.cornell_camera(64, 64)
    hip_ctx.trace_timing(reset=True)
    assert hip_ctx.trace_timing() == (0.0, 0)
    total_ms, total_n = 0.0, 0
    for spp in (2, 4):
        _, st = g.render(cam, 64, 64, spp, max_depth=4, seed=1)
        assert st["trace_launches"] >= 2 and st["trace_ms"] > 0.0
        total_ms, total_n = total_ms + st["trace_ms"], total_n + st["trace_launches"]
        ms, n = hip_ctx.trace_timing()
        assert n == total_n and abs(ms - total_ms) <= 1e-9 * total_ms
    assert hip_ctx.trace_timing(reset=True)[1] == total_n        # the read that resets still returns the totals
    assert hip_ctx.trace_timing() == (0.0, 0)
    g.close()
