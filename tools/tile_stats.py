"""Work-list statistics along a run: python tools/tile_stats.py [cols=1024] [steps ...]"""
import sys
sys.path.insert(0, ".")
import torch  # noqa
from sph_amd import scenes
from sph_amd.solver import Solver
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
marks = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 10, 500, 1000, 1010]
p, a = scenes.dam_break(cols)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    done = 0
    for m in marks:
        s.stepSimulation(nsteps=m - done)
        done = m
        print("step %5d: %s" % (m, s.tile_stats()), flush=True)
