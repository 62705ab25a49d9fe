"""N > 1 path on CPU: two processes (gloo) each take their tile share from the library's host-side
tile partition (pbrt_hip_tile_partition, the function pbrt_hip_render uses), render those tiles —
here with the CPU oracle standing in for the GPU kernels — into a zero-initialised full-size film,
and reduce(SUM) to rank 0 exactly as bench.py does over RCCL. The sum must equal the one-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, BOUNDS = 72, 52, 2, (3, 2, 70, 49)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle
    import pbrt_hip
    from pbrt_hip import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = oracle.OracleScene(scenes.cornell_box())
    cam = scenes.camera_dict_to_floats(scenes.cornell_camera(W, H))
    film = np.zeros((H, W, 4), dtype=np.float32)
    for x, y in pbrt_hip.tile_partition(BOUNDS, rank, world):
        b = (int(x), int(y), min(int(x) + 16, BOUNDS[2]), min(int(y) + 16, BOUNDS[3]))
        tile, _ = osc.render(cam, W, H, SPP, max_depth=8, seed=4, bounds=b, n_threads=1)
        film += tile
    mine = torch.tensor([float((film[..., 3] > 0).sum())])   # pixels this rank owns
    t = torch.from_numpy(film)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)   # the film reduce (RCCL on the GPUs)
    dist.all_reduce(mine)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().reshape(-1), mine.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_sharding_and_film_reduce(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from pbrt_hip import scenes
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    film, n_owned = got[:-1].reshape(H, W, 4), got[-1]
    osc = oracle.OracleScene(scenes.cornell_box())
    ref, _ = osc.render(scenes.camera_dict_to_floats(scenes.cornell_camera(W, H)), W, H, SPP, max_depth=8, seed=4,
                        bounds=BOUNDS)
    osc.close()
    assert n_owned == (BOUNDS[2] - BOUNDS[0]) * (BOUNDS[3] - BOUNDS[1])   # every pixel owned by exactly one rank
    assert film.tobytes() == ref.tobytes()
