import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import orc
from sph_amd.solver import Solver
O = orc.Oracle(); P = O.default_params()
def G(name): return np.load(os.path.join(orc.GOLDEN, name + ".npz"), allow_pickle=False)
s = Solver(P, n_max=1 << 16)
for nm, key in (("f4_phases_n1024", "after_density"), ("f5_dense_block", "after_density"), ("f9_invalid", "state")):
    a = G(nm)[key]
    s.upload(a); s.computeForces((0.0, 0.0), 0.0)
    got = s.download(); want = O.forces_cells(P, a.copy(), (0.0, 0.0), 0.0)
    bad = np.nonzero((got["vel"].view(np.uint32) != want["vel"].view(np.uint32)).any(1))[0]
    print(nm, "n", len(a), "bad", len(bad))
    for i in bad[:5]:
        print("  ", i, got["vel"][i], want["vel"][i], a["pos"][i], a["density"][i], a["pressure"][i], a["valid"][i])
