"""What the in-cell ranking costs where cells are crowded: phase times (HIP events) of
  (a) the reference's own box at N = 65536 (about 390 particles per cell),
  (b) a 'collapsed' scene: K particles inside one cell of the reference box plus a 4096-particle lattice.
    python tools/rank_cost.py [K=20000]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.params import PARTICLE_DTYPE, SphParams  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402


def collapsed(k):
    base = scenes.lattice(4096)
    a = np.zeros(len(base) + k, PARTICLE_DTYPE)
    a[: len(base)] = base
    rng = np.random.default_rng(5)
    blob = a[len(base):]
    blob["pos"][:, 0] = (1.0 + 0.1 * rng.random(k)).astype(np.float32)  # inside one 0.15-wide cell
    blob["pos"][:, 1] = (1.0 + 0.1 * rng.random(k)).astype(np.float32)
    blob["density"] = 1000.0
    blob["valid"] = 1
    return a[rng.permutation(len(a))].copy()


def run(name, a, steps):
    p = SphParams.reference()
    with Solver(p, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=1)
        s.synchronize()
        s.set_profiling(True)
        s.reset_phase_times()
        t0 = time.perf_counter()
        s.stepSimulation(nsteps=steps)
        s.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ph = s.phase_times()
        s.set_profiling(False)
        s.buildNeighbors()
        _, _, cs, ce = s.cells()
        print("%s: N=%d, largest cell %d, %.2f ms/step; phases (ms/step): %s" % (
            name, len(a), int((ce - cs).max()), dt * 1e3, ", ".join("%s %.3f" % (k, v[0] / steps) for k, v in ph.items() if v[1])), flush=True)


k = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
run("reference box", scenes.lattice(65536), 5)
run("collapsed", collapsed(k), 2)
