"""Long dam-break run: speed and sanity over time (collapse, splash, sloshing)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from sph_amd import scenes
from sph_amd.solver import Solver
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
params, a = scenes.dam_break(cols)
with Solver(params, n_max=len(a)) as s:
    s.upload(a)
    for k in range(chunks):
        t0 = time.perf_counter(); s.stepSimulation(nsteps=per); s.synchronize(); dt = time.perf_counter() - t0
        s.computeDensityPressure()
        nc = s.neighbor_counts()
        g = s.download()
        print("steps %6d  ms/step %.4f  p-u/s %.3e  nan %d  neigh mean %.1f max %d  x-extent %.1f  vmax %.2f" % (
            (k + 1) * per, dt / per * 1e3, len(a) * per / dt, int(np.isnan(g["pos"]).any(1).sum()), nc.mean(), nc.max(),
            g["pos"][:, 0].max(), np.abs(g["vel"]).max()), flush=True)
