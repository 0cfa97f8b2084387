"""Debug helper: one failing slab fuzz seed, rank-by-rank column ranges around the first steps."""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "debug"))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
import fuzz_slab
from sph_amd.slab import make_hip_slab, grid_column


def worker(rank, world, port, seed):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params, a, _, plan = fuzz_slab.scene(seed)
        slab = make_hip_slab(params, a, device=0)
        drv = slab.backend.drv
        print(rank, "bounds", slab.bounds, "nx", drv.nx, "n_max", slab.backend.solver.n_max, "cap", slab.backend.cap, flush=True)
        for step in range(6):
            try:
                slab.step(1)
                n = slab.check()
                got, ids = slab.backend.solver.download(with_ids=True)
                col = grid_column(got["pos"][:, 0], params, drv)
                print(rank, "step", step, "n", n, "cols", int(col.min()), int(col.max()), "nan", int(np.isnan(got["pos"]).any()), flush=True)
            except Exception as e:
                print(rank, "step", step, "ERR", str(e)[:200], flush=True)
                break
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    seed = int(sys.argv[1]); world = int(sys.argv[2])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, seed), nprocs=world, join=True)
