"""Test infrastructure: the C++ slab driver (S slabs on one GPU, copy transport) against the single-GPU run over a long
horizon with periodic rebalances, bit for bit:  python tests/debug/soak_mg.py [cols=512] [slabs=4] [steps=1000] [every=50]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from sph_amd import scenes
from sph_amd.mg import MultiGpuSolver
from sph_amd.solver import Solver

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
every = int(sys.argv[4]) if len(sys.argv) > 4 else 50
params, a = scenes.dam_break(cols)
a["vel"][:, 0] = 2.0
bad = 0
with Solver(params, n_max=len(a)) as s, MultiGpuSolver(params, a, devices=[0] * world) as m:
    s.upload(a)
    done = 0
    while done < steps:
        k = min(250, steps - done)
        s.stepSimulation(nsteps=k)
        m.step(k, rebalance_every=every)
        m.synchronize()
        done += k
        same = orc.same_bits_nan(m.download(), s.download())
        bad += 0 if same else 1
        print("steps %5d: %s  counts %s bounds %s rebalances %d" % (done, "bit-exact" if same else "MISMATCH", m.counts(), m.bounds(), m.rebalances), flush=True)
sys.exit(1 if bad else 0)
