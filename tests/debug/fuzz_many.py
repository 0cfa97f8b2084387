"""Test infrastructure (uses the oracle): runs the fuzz tests of tests/test_gpu_parity.py for a range of extra seeds on the GPU box:
    python tests/debug/fuzz_many.py <first seed> <last seed + 1> [box|scaled]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T

import numpy as np
import orc
from sph_amd import scenes
from sph_amd.solver import Solver


def fuzz_scaled(seed):
    """Scaled scenes (dam break / box miniatures) with jitter and random velocities, many steps in one call
    (graph replay for small n), against the oracle."""
    rng = np.random.default_rng(5000 + seed)
    cols = int(rng.choice([8, 20, 33, 64, 100, 150]))
    params, a = scenes.dam_break(cols, int(rng.integers(4, cols + 1))) if seed % 2 else scenes.uniform_box(cols)
    a["pos"] += rng.uniform(-0.01, 0.01, size=a["pos"].shape).astype(np.float32)
    a["vel"] = rng.normal(0.0, float(rng.choice([0.0, 0.5, 3.0])), size=a["vel"].shape).astype(np.float32)
    steps = int(rng.choice([1, 7, 18, 27, 40]))
    want = T.O.step_cells(params, a.copy(), steps)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=steps)
        got = s.download()
    assert orc.same_bits_nan(got, want), (seed, cols, steps, len(a))


lo, hi = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "box"
fn = T.test_random_scenes_fuzz if which == "box" else fuzz_scaled
bad = []
t0 = time.time()
for seed in range(lo, hi):
    try:
        fn(seed)
    except AssertionError:
        bad.append(seed)
        print("MISMATCH seed", seed, flush=True)
    if seed % 50 == 0:
        print("seed", seed, "%.0f s" % (time.time() - t0), flush=True)
print("done", lo, hi, "mismatches:", bad)
sys.exit(1 if bad else 0)
