"""Step time of one slab of an 8-rank weak-scaling scene (global grid 8x wider), with the cell table
restricted to the slab's columns vs maintained for the whole grid."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from sph_amd import scenes, lib as _lib
from sph_amd.solver import Solver
from sph_amd.slab import balanced_bounds
params, n_total, hist_fn, make_local = scenes.dam_break_sharded(1024 * 8, 1024)
drv = _lib.derive(params)
hist = hist_fn(drv)
b = balanced_bounds(hist, 8)
a, ids = make_local(b[0], b[1], drv)
print("grid", drv.nx, drv.ny, "slab", b[0], b[1], "particles", len(a))
for owned in ((b[0], b[1]), (0, drv.nx)):
    with Solver(params, n_max=len(a) + 1024) as s:
        s.upload(a)
        s.set_owned_columns(*owned)
        s.stepSimulation(nsteps=5); s.synchronize()
        t0 = time.perf_counter(); s.stepSimulation(nsteps=40); s.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print("owned", owned, "ms/step %.4f" % (dt * 1e3))
