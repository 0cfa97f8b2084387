"""Phase breakdown of the reference's own fixed-box lattice scene at a given N (dense cells)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from sph_amd import scenes, lib
from sph_amd.solver import Solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a = scenes.lattice(n)
s = Solver(None, n_max=len(a), device=0)
s.upload(a)
s.stepSimulation(nsteps=int(sys.argv[3]) if len(sys.argv) > 3 else 3)
torch.cuda.synchronize()
s.set_profiling(True)
s.reset_phase_times()
s.stepSimulation(nsteps=steps)
torch.cuda.synchronize()
for k, v in s.phase_times().items():
    print("%-10s %.4f ms/step" % (k, v[0] / steps))
s.buildNeighbors()
keys, order, cs, ce = s.cells()
if True:
    occ = ce - cs
    print("cells", len(occ), "max occupancy", occ.max(), "mean nonempty", occ[occ > 0].mean())
