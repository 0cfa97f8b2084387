import sys
sys.path.insert(0, ".")
import numpy as np, torch
from sph_amd import scenes
from sph_amd.params import SphParams
from sph_amd.solver import Solver
p = SphParams.reference(); a = scenes.lattice(65536)
with Solver(p, n_max=len(a)) as s:
    s.upload(a); s.stepSimulation(nsteps=1000); b = s.download()
    s.computeDensityPressure(); cnt = s.neighbor_counts(); keys, order, cs, ce = s.cells()
nan = np.isnan(b["pos"]).any(axis=1)
print("NaN pos", nan.sum(), "NaN vel", np.isnan(b["vel"]).any(axis=1).sum(), "inf pos", np.isinf(b["pos"]).any(axis=1).sum())
pop = ce - cs
big = np.argsort(pop)[-5:]
ny = 13
for k in big: print("cell", k, "col", k // ny, "row", k % ny, "pop", pop[k])
k = big[-1]
ids = order[cs[k]:ce[k]]
q = b[ids]
print("in biggest: NaN", np.isnan(q["pos"]).any(axis=1).sum(), "x range", np.nanmin(q["pos"][:,0]), np.nanmax(q["pos"][:,0]), "y range", np.nanmin(q["pos"][:,1]), np.nanmax(q["pos"][:,1]))
print("neighbour counts in biggest: mean", cnt[ids].mean(), "max", cnt[ids].max(), "zeros", (cnt[ids]==0).sum())
u, c = np.unique(q["pos"], axis=0, return_counts=True)
print("distinct positions", len(u), "most common", u[np.argmax(c)], c.max())
