"""Cost of one SlabSolver.rebalance() on a one-rank group (nccl), section by section, and of the steps right after it.
    python tools/rebalance_cost.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import torch.distributed as dist
from sph_amd import lib as _lib, scenes
from sph_amd.slab import make_hip_slab

os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("BACKEND", "nccl"), device_id=torch.device("cuda", 0))
params, n_total, hist_fn, make_local = scenes.dam_break_sharded(1024, 1024)
drv = _lib.derive(params)
slab = make_hip_slab(params, make_local, 0, hist=hist_fn(drv))
sync = torch.cuda.synchronize
slab.step(20); sync()
def timed(f):
    sync(); t = time.perf_counter(); r = f(); sync(); return (time.perf_counter() - t) * 1e3, r
for rep in range(3):
    t_steps, _ = timed(lambda: slab.step(50))
    b = slab.backend
    t_drop, _ = timed(b.drop_ghosts)
    t_hist, h = timed(b.column_histogram)
    def allred():
        t = torch.from_numpy(h.astype(np.int64)).to(b.device); dist.all_reduce(t); return t.cpu().numpy()
    t_all, _ = timed(allred)
    t_reb, _ = timed(slab.rebalance)
    t_after, _ = timed(lambda: slab.step(1))
    t_after2, _ = timed(lambda: slab.step(1))
    print("50 steps %.2f ms (%.4f/step); drop_ghosts %.3f, histogram %.3f, all_reduce %.3f; whole rebalance() %.3f ms; first step after %.3f, second %.3f ms" % (
        t_steps, t_steps / 50, t_drop, t_hist, t_all, t_reb, t_after, t_after2), flush=True)
dist.destroy_process_group()
