"""Test infrastructure (uses the oracle): a long full-size run against the oracle, bit for bit, in chunks:
    python tests/debug/soak_parity.py [cols=512] [chunks=6] [steps per chunk=250]
The HIP state is compared with the oracle's after every chunk (both advance from their own state)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from sph_amd import scenes
from sph_amd.solver import Solver

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
per = int(sys.argv[3]) if len(sys.argv) > 3 else 250
O = orc.Oracle()
params, a = scenes.dam_break(cols)
want = a.copy()
bad = 0
with Solver(params, n_max=len(a)) as s:
    s.upload(a)
    for k in range(chunks):
        t0 = time.time()
        s.stepSimulation(nsteps=per)
        got = s.download()
        st = s.tile_stats()
        t1 = time.time()
        O.step_cells(params, want, per)
        same = orc.same_bits_nan(got, want)
        bad += 0 if same else 1
        print("steps %5d: %s  (hip %.1f s, oracle %.1f s)  tiles %d long %d dense %d/%d  nan %d  vmax %.2f" % (
            (k + 1) * per, "bit-exact" if same else "MISMATCH", t1 - t0, time.time() - t1, st["tiles"], st["long"], st["dense_density"],
            st["dense_force"], int(np.isnan(got["pos"]).any(1).sum()), float(np.nanmax(np.abs(got["vel"])))), flush=True)
sys.exit(1 if bad else 0)
