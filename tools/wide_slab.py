"""The last slab of an 8-rank weak-scaling dam break on one GPU (no neighbours: timing of the grid build over a wide, mostly
empty column range, not physics):  python tools/wide_slab.py [world=8] [steps=60]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa
from sph_amd import scenes, lib as _lib
from sph_amd.solver import Solver

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
params, n_total, hist_fn, make_local = scenes.dam_break_sharded(1024 * world, 1024)
drv = _lib.derive(params)
h = hist_fn(drv)
cum = np.cumsum(h)
c0 = int(np.searchsorted(cum, n_total * (world - 1) / world)) + 1
a, ids = make_local(c0, drv.nx, drv)
with Solver(params, n_max=2 * len(a)) as s:
    s.set_owned_columns(c0, drv.nx)
    s.upload(a)
    s.stepSimulation(nsteps=5)
    s.synchronize()
    t0 = time.perf_counter()
    s.stepSimulation(nsteps=steps)
    s.synchronize()
    dt = (time.perf_counter() - t0) / steps
    s.set_profiling(True)
    s.reset_phase_times()
    s.stepSimulation(nsteps=5)
    s.synchronize()
    ph = s.phase_times()
    print("slab [%d, %d) of %d columns, %d particles: %.4f ms/step; phases (us/step): %s" % (
        c0, drv.nx, drv.nx, len(a), dt * 1e3, {k: round(v[0] / 5 * 1e3, 1) for k, v in ph.items()}), flush=True)
