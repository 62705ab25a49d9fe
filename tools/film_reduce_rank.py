"""One rank of the C-ABI multi-GPU path: render this rank's tiles into a device film, pbrt_hip_film_reduce onto
rank 0, rank 0 compares with the full frame rendered alone. The communicator id travels through a file.
Usage: film_reduce_rank.py RANK WORLD DEVICE ID_FILE OUT_NPZ   (tests/test_gpu_film_reduce.py starts the ranks)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
import numpy as np
import torch
import pbrt_hip
from pbrt_hip import scenes

rank, world, device, id_file, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
ctx = pbrt_hip.Context(device)
if rank == 0:
    uid = pbrt_hip.comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            sys.exit("no communicator id after 60 s")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
comm = pbrt_hip.Comm(ctx, world, rank, uid)
W, H, spp = 96, 64, 8
sc = pbrt_hip.Scene(ctx, scenes.cornell_box())
cam = scenes.cornell_camera(W, H)
film = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{device}")
sc.render(cam, W, H, spp, max_depth=3, seed=5, tile_rank=rank, tile_world=world, d_film_ptr=film.data_ptr())
torch.cuda.synchronize()
comm.film_reduce(film.data_ptr(), W * H, root=0)
if rank == 0:
    full, _ = sc.render(cam, W, H, spp, max_depth=3, seed=5)
    np.savez(out, reduced=film.cpu().numpy(), full=full)
comm.close()
sc.close()
ctx.close()
print(f"rank {rank} done", flush=True)
