"""N > 1 path on CPU: two and four processes (gloo) each take their tile share from the library's host-side
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


def _bench_worker(rank, world, port, out_path, fail):
    """bench.py's own measured loop (run_steps) and stage discipline (Stage) on two gloo ranks, the CPU oracle standing in
    for pbrt_hip_render_device: each step renders this rank's tile share into a zeroed film and reduces it to rank 0."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import bench
    import oracle
    import pbrt_hip
    from pbrt_hip import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if fail:
        os.environ["PBRT_BENCH_FAIL"] = fail
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = oracle.OracleScene(scenes.cornell_box())
    cam = scenes.camera_dict_to_floats(scenes.cornell_camera(W, H))
    tiles = pbrt_hip.tile_partition(BOUNDS, rank, world)

    def render_into(film):
        film.zero_()   # as pbrt_hip_render_device does
        rays = 0
        for x, y in tiles:
            b = (int(x), int(y), min(int(x) + 16, BOUNDS[2]), min(int(y) + 16, BOUNDS[3]))
            tile, st = osc.render(cam, W, H, SPP, max_depth=8, seed=4, bounds=b, n_threads=1)
            film += torch.from_numpy(tile)
            rays += st["rays"]
        return {"rays": rays}

    films = [torch.zeros((H, W, 4), dtype=torch.float32) for _ in range(2)]
    stage = bench.Stage(dist, torch, torch.device("cpu"), True, rank)
    try:
        seconds, stats, last = bench.run_steps(render_into, films, 2, 1, dist, True, lambda: None, None, stage)
        rays = stage.collective("sum over ranks", lambda: _sum(dist, torch, sum(st["rays"] for st in stats)))
        # bench.py's per-rank report gather (config.ranks): all_gather_object over gloo, in rank order, the same on every rank
        mine = {"rank": rank, "uuid": f"fake-device-{rank}", "pci_bus_id": "", "render_ms_per_step": {"mean": 10.0 * (rank + 1), "max": 12.0 * (rank + 1)},
                "rays_per_step": sum(st["rays"] for st in stats) // 2, "pid": os.getpid()}
        reports = stage.collective("gather rank reports", lambda: bench.gather_rank_reports(dist, True, world, mine))
        assert [r["rank"] for r in reports] == list(range(world)) and reports[rank] == mine
        assert bench.check_distinct_devices(reports, one_gpu=False) == world
        assert bench.load_balance(reports) == pytest.approx(20.0 / 15.0, abs=1e-3)
        same = [dict(r, uuid="one-device") for r in reports]
        assert bench.check_distinct_devices(same, one_gpu=True) == 1
        with pytest.raises(RuntimeError, match="distinct device"):
            bench.check_distinct_devices(same, one_gpu=False)
        assert sum(r["rays_per_step"] for r in reports) == rays // 2
    except bench.StageFailed:
        os._exit(4)   # what bench.py's main() does: leave, non-zero, without waiting for the others
    if rank == 0:
        np.save(out_path, np.concatenate([last.numpy().reshape(-1), [rays / 2, seconds]]))
    dist.barrier()
    dist.destroy_process_group()


def _sum(dist, torch, v):
    t = torch.tensor([float(v)], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_bench_loop_on_two_ranks(tmp_path):
    """bench.run_steps + bench.Stage with world 2: the reduced film of the last step is the one-process frame, byte for
    byte, and the rays of the ranks add up to the frame's."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from pbrt_hip import scenes
    out = str(tmp_path / "bench_film.npy")
    mp.spawn(_bench_worker, args=(2, _free_port(), out, ""), nprocs=2, join=True)
    got = np.load(out)
    film, rays = got[:-2].reshape(H, W, 4), got[-2]
    osc = oracle.OracleScene(scenes.cornell_box())
    ref, st = osc.render(scenes.camera_dict_to_floats(scenes.cornell_camera(W, H)), W, H, SPP, max_depth=8, seed=4, bounds=BOUNDS)
    osc.close()
    assert film.astype(np.float32).tobytes() == ref.tobytes()
    assert rays == st["rays"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail", ["render@1", "sum over ranks@0"])
def test_bench_stage_failure_ends_every_rank_nonzero(tmp_path, fail):
    """A rank that fails inside a stage is seen by the others at the end of that stage: nobody waits in the next
    collective, every rank exits non-zero (a hang or mismatch must not end in rc 0)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, str(tmp_path / "x.npy"), fail)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(not p.is_alive() for p in procs), "a rank is still waiting for the one that failed"
    assert [p.exitcode for p in procs] == [4, 4]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])   # from four ranks on the Morton deal is a 2-D lattice, not tile columns
def test_tile_sharding_and_film_reduce(tmp_path, world):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from pbrt_hip import scenes
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    film, n_owned = got[:-1].reshape(H, W, 4), got[-1]
    osc = oracle.OracleScene(scenes.cornell_box())
    ref, _ = osc.render(scenes.camera_dict_to_floats(scenes.cornell_camera(W, H)), W, H, SPP, max_depth=8, seed=4,
                        bounds=BOUNDS)
    osc.close()
    assert n_owned == (BOUNDS[2] - BOUNDS[0]) * (BOUNDS[3] - BOUNDS[1])   # every pixel owned by exactly one rank
    assert film.tobytes() == ref.tobytes()


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_morton_deal_spreads_every_rank_over_the_frame_in_both_directions(w, h):
    """the 16x16 tiles are dealt round-robin in MORTON order. With 120 (1080p) or 240 (4K) tiles per row the
    row-major deal gives a rank whole tile COLUMNS at N = 2, 4, 8 (16-px stripes every 16 N px); the Morton deal gives every
    rank a 2-D lattice from N = 4 on (N = 4: every other tile of every other row, N = 8: a 4 x 2 lattice). At N = 2 the low bit
    of a Z-order index IS the column parity, so the two deals coincide there — stated, not hidden. Either way the shares are
    a partition of the frame into equal counts (+-1)."""
    import pbrt_hip
    bounds = (0, 0, w, h)
    ntx, nty = (w + 15) // 16, (h + 15) // 16
    for world in (2, 4, 8):
        for order in (pbrt_hip.TILE_ORDER_MORTON, pbrt_hip.TILE_ORDER_ROW_MAJOR):
            shares = [pbrt_hip.tile_partition(bounds, r, world, order) // 16 for r in range(world)]
            owner = np.full((nty, ntx), -1)
            for r, t in enumerate(shares):
                assert (owner[t[:, 1], t[:, 0]] == -1).all()
                owner[t[:, 1], t[:, 0]] = r
            assert (owner >= 0).all() and max(len(t) for t in shares) - min(len(t) for t in shares) <= 1
            whole_columns = all((owner[:, x] == owner[0, x]).all() for x in range(ntx))
            if order == pbrt_hip.TILE_ORDER_ROW_MAJOR or world == 2:
                assert whole_columns            # what the Morton deal is there to avoid (N = 2: inherent to a mod-2 deal of a Z curve)
            else:
                assert not whole_columns
                for r, t in enumerate(shares):
                    # every rank has tiles in (nearly) every tile row and in every second tile column at least: no stripes
                    assert len(set(t[:, 1].tolist())) >= nty // 2 and len(set(t[:, 0].tolist())) >= ntx // 4
                    # ... and any 4 x 4 block of tiles inside the frame holds a tile of every rank (N <= 8)
                for y in range(0, nty - 3, 4):
                    for x in range(0, ntx - 3, 4):
                        assert len(set(owner[y:y + 4, x:x + 4].ravel().tolist())) == world
