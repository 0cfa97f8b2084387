"""Measures the host-side enqueue time of one step (no sync inside) vs the GPU time."""
import sys, time
sys.path.insert(0, ".")
import torch
from sph_amd import scenes
from sph_amd.solver import Solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p, a = scenes.dam_break(n)
s = Solver(p, n_max=len(a)); s.upload(a)
s.stepSimulation(nsteps=5); s.synchronize()
for prof in (False, True):
    s.set_profiling(prof)
    t0 = time.perf_counter(); s.stepSimulation(nsteps=100); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
    print("N=%d profiling=%s: host enqueue %.1f us/step, total %.1f us/step" % (len(a), prof, (t1 - t0) * 1e4, (t2 - t0) * 1e4))
