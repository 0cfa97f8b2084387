"""Test infrastructure (uses the oracle): random solver constants (h, dt, mass, viscosity, stiffness, rest
density, damping, max velocity, sigma, epsilon, gravity, force cap) on random lattices, HIP path against
the oracle bit for bit:   python tests/debug/fuzz_params.py <first seed> <last seed + 1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from sph_amd import scenes
from sph_amd.params import SphParams
from sph_amd.solver import Solver

O = orc.Oracle()


def one(seed):
    rng = np.random.default_rng(20000 + seed)
    base = SphParams.reference()
    p = base.copy(h=rng.choice([0.04, 0.1, 0.15, 0.22, 0.4]), dt=rng.choice([1e-4, 5e-4, 2e-3]),
                  mass=rng.choice([0.01, 0.05, 0.3]), viscosity=rng.choice([0.0, 0.3, 2.0]),
                  stiffness=rng.choice([50.0, 1000.0, 2e4]), rest_density=rng.choice([10.0, 1000.0, 3000.0]),
                  damping=rng.choice([0.5, 0.9, 1.0]), max_vel=rng.choice([1.0, 5.0, 50.0]),
                  sigma=rng.choice([0.01, 0.05, 0.12]), epsilon=rng.choice([0.1, 1.0, 10.0]),
                  gravity=rng.choice([0.0, 10.0]), force_cap=rng.choice([100.0, 1000.0, 1e5]))
    n = int(rng.choice([100, 1024, 3000, 6000]))
    a = scenes.lattice(n)
    a["pos"] += rng.uniform(-0.004, 0.004, size=a["pos"].shape).astype(np.float32)
    a["vel"] = rng.normal(0.0, 0.5, size=a["vel"].shape).astype(np.float32)
    steps = int(rng.choice([1, 4, 20]))
    mouse, strength = ((0.7, 1.1), -30000.0) if seed % 3 == 0 else ((0.0, 0.0), 0.0)
    want = O.step_cells(p, a.copy(), steps, mouse=mouse, strength=strength)
    with Solver(p, n_max=n) as s:
        s.upload(a)
        s.stepSimulation(mouse, strength, steps)
        got = s.download()
    return orc.same_bits_nan(got, want), (float(p.h), n, steps)


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        ok, info = one(seed)
        if not ok:
            bad.append(seed)
            print("MISMATCH seed", seed, info, flush=True)
        if seed % 25 == 0:
            print("seed", seed, flush=True)
    print("done", lo, hi, "mismatches:", bad)
    sys.exit(1 if bad else 0)
