"""pbrt_hip_comm_* / pbrt_hip_film_reduce on the one GPU of the test box: a one-rank communicator (RCCL comes up,
the in-place reduce and all-reduce leave the film as it is) and argument checks. More ranks need more GPUs:
tools/film_reduce_rank.py is the per-rank program for such a node, bench.py --gpus N runs the same check."""
import ctypes

import numpy as np
import pytest

import pbrt_hip
from pbrt_hip import scenes

pytestmark = pytest.mark.gpu


class DeviceFilm:
    """A device buffer from the HIP runtime libpbrt_hip.so itself runs on (no second ROCm stack in the process)."""

    def __init__(self, n_floats):
        self.hip = ctypes.CDLL("libamdhip64.so.7")
        self.hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self.hip.hipFree.argtypes = [ctypes.c_void_p]
        self.n = n_floats
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), 4 * n_floats) == 0
        self.ptr = p.value
        assert self.hip.hipMemset(self.ptr, 0, 4 * n_floats) == 0

    def numpy(self, shape):
        out = np.zeros(self.n, dtype=np.float32)
        assert self.hip.hipMemcpy(out.ctypes.data, self.ptr, 4 * self.n, 2) == 0   # device to host, synchronous
        return out.reshape(shape)

    def free(self):
        self.hip.hipFree(self.ptr)


def test_single_rank_film_reduce_is_the_identity(hip_ctx):
    uid = pbrt_hip.comm_unique_id()
    assert len(uid) == pbrt_hip.COMM_ID_BYTES and any(uid)
    comm = pbrt_hip.Comm(hip_ctx, 1, 0, uid)
    w, h = 80, 48
    sc = pbrt_hip.Scene(hip_ctx, scenes.cornell_box())
    cam = scenes.cornell_camera(w, h)
    host, _ = sc.render(cam, w, h, 4, max_depth=3, seed=2)
    film = DeviceFilm(w * h * 4)
    sc.render(cam, w, h, 4, max_depth=3, seed=2, d_film_ptr=film.ptr)
    assert np.array_equal(film.numpy((h, w, 4)), host)
    comm.film_reduce(film.ptr, w * h, root=0)      # ncclReduce, one rank
    assert np.array_equal(film.numpy((h, w, 4)), host)
    comm.film_reduce(film.ptr, w * h, root=-1)     # ncclAllReduce
    assert np.array_equal(film.numpy((h, w, 4)), host)
    comm.film_reduce(film.ptr, 0, root=0)          # empty film
    with pytest.raises(pbrt_hip.PbrtHipError):
        comm.film_reduce(film.ptr, w * h, root=1)  # root outside the communicator
    with pytest.raises(pbrt_hip.PbrtHipError):
        comm.film_reduce(0, w * h, root=0)         # null film
    film.free()
    comm.close()
    sc.close()


def test_comm_create_rejects_bad_ranks(hip_ctx):
    uid = pbrt_hip.comm_unique_id()
    for world, rank in ((0, 0), (2, 2), (2, -1)):
        with pytest.raises(pbrt_hip.PbrtHipError):
            pbrt_hip.Comm(hip_ctx, world, rank, uid)
    with pytest.raises(ValueError):
        pbrt_hip.Comm(hip_ctx, 1, 0, uid[:64])


def test_the_abis_own_device_film(hip_ctx):
    """pbrt_hip_film_create / _download / _destroy: the film a host WITHOUT a device allocator renders into and merges (a Rust or
    C caller of include/pbrt_hip.h). Zero-filled; two tile shares rendered into two such films and added = the frame; the one-rank
    reduce leaves it; bad arguments are errors."""
    w, h = 96, 64
    sc = pbrt_hip.Scene(hip_ctx, scenes.cornell_box())
    cam = scenes.cornell_camera(w, h)
    host, _ = sc.render(cam, w, h, 4, max_depth=3, seed=5)
    film = pbrt_hip.DeviceFilm(hip_ctx, w, h)
    assert not film.download().any()
    shares = []
    for rank in range(2):
        sc.render(cam, w, h, 4, max_depth=3, seed=5, tile_rank=rank, tile_world=2, d_film_ptr=film.ptr)
        shares.append(film.download())
    assert np.array_equal(shares[0] + shares[1], host) and shares[0].any() and shares[1].any()
    sc.render(cam, w, h, 4, max_depth=3, seed=5, d_film_ptr=film.ptr)
    comm = pbrt_hip.Comm(hip_ctx, 1, 0, pbrt_hip.comm_unique_id())
    comm.film_reduce(film.ptr, w * h, root=0)
    assert np.array_equal(film.download(), host)
    comm.close()
    L = pbrt_hip.lib()
    out = ctypes.c_void_p()
    assert L.pbrt_hip_film_create(hip_ctx.h, 0, ctypes.byref(out)) == 1 and L.pbrt_hip_film_create(hip_ctx.h, 16, None) == 1
    assert L.pbrt_hip_film_download(hip_ctx.h, None, 16, None) == 1
    L.pbrt_hip_film_destroy(hip_ctx.h, None)       # a no-op
    film.close()
    film.close()                                   # idempotent
    sc.close()
