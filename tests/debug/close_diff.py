import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import orc
from sph_amd.solver import Solver
from sph_amd.params import PARTICLE_DTYPE
O = orc.Oracle(); P = O.default_params()
rng = np.random.default_rng(77)
n = 3000
a = np.zeros(n, PARTICLE_DTYPE)
a["pos"] = rng.uniform(0.3, 1.5, (n, 2)).astype(np.float32)
a["vel"] = rng.normal(0, 0.3, (n, 2)).astype(np.float32)
a["density"] = 1000.0
a["valid"] = rng.random(n) >= 0.0
twins = a[:600].copy()
off = np.array([0.0, 1e-9, 1e-8, 1e-7, 5e-7, 9.9e-7, 1e-6, 1.1e-6, 2e-6, 1e-5], np.float32)
twins["pos"][:, 0] += off[np.arange(600) % len(off)]
twins["pos"][300:, 1] += off[(np.arange(300) // 3) % len(off)]
both = np.zeros(len(a) + len(twins), PARTICLE_DTYPE); both[:n] = a; both[n:] = twins; a = both
s = Solver(P, n_max=1 << 16)
steps = int(os.environ.get("STEPS", "1"))
s.upload(a); s.stepSimulation(nsteps=steps)
got = s.download(); want = O.step_cells(P, a.copy(), steps)
for f in ("pos", "vel"):
    bad = np.nonzero((got[f].view(np.uint32) != want[f].view(np.uint32)).any(1))[0]
    print(f, "bad", len(bad), bad[:20])
    for i in bad[:6]:
        k = (i - n) % len(off) if i >= n else (i % len(off) if i < 600 else -1)
        print("  ", i, "off", off[k] if k >= 0 else None, got[f][i], want[f][i], "rho", got["density"][i], want["density"][i])
print("density bad", (got["density"].view(np.uint32) != want["density"].view(np.uint32)).sum())
