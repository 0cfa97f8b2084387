"""Phase times of the reference's own box at a crowded N (default 65536) along the run (perf-test protocol: 1000 steps)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.params import SphParams  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = SphParams.reference()
a = scenes.lattice(n)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    done = 0
    for upto in (10, 100, 300, 600, 1000):
        s.set_profiling(False)
        s.stepSimulation(nsteps=upto - done - 5)
        s.synchronize()
        s.set_profiling(True)
        s.reset_phase_times()
        t0 = time.perf_counter()
        s.stepSimulation(nsteps=5)
        s.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ph = s.phase_times()
        done = upto
        s.set_profiling(False)
        s.computeDensityPressure()
        cnt = s.neighbor_counts()
        _, _, cs, ce = s.cells()
        print("step %4d: %.2f ms/step; largest cell %d, neighbours mean %.0f max %d; phases (ms): %s" % (
            upto, dt * 1e3, int((ce - cs).max()), cnt.mean(), cnt.max(),
            ", ".join("%s %.3f" % (k, v[0] / 5) for k, v in ph.items() if v[1])), flush=True)
