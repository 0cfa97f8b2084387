"""Test infrastructure: one scene of the scaled fuzz (tests/test_gpu_parity.py::test_scaled_scenes_fuzz) in detail --
which fields of which particles differ from the oracle:  python tests/debug/scaled_seed.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from sph_amd import scenes
from sph_amd.solver import Solver
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
cols = int(rng.choice([8, 20, 33, 64, 100, 150]))
params, a = scenes.dam_break(cols, int(rng.integers(4, cols + 1))) if seed % 2 else scenes.uniform_box(cols)
a["pos"] += rng.uniform(-0.01, 0.01, size=a["pos"].shape).astype(np.float32)
a["vel"] = rng.normal(0.0, float(rng.choice([0.0, 0.5, 3.0])), size=a["vel"].shape).astype(np.float32)
steps = int(rng.choice([1, 7, 18, 27, 40]))
O = orc.Oracle()
want = O.step_cells(params, a.copy(), steps)
with Solver(params, n_max=len(a)) as s:
    s.upload(a); s.stepSimulation(nsteps=steps); got = s.download()
    s.upload(a); s.computeDensityPressure(); gd = s.download()
wd, _ = O.density_cells(params, a.copy())
for f in ("pos", "vel", "density", "pressure"):
    x, y = got[f].view(np.uint32).reshape(len(a), -1), want[f].view(np.uint32).reshape(len(a), -1)
    bad = np.nonzero((x != y).any(axis=1))[0]
    print(f, "differs for", len(bad), "particles", bad[:8], got[f][bad[:3]], want[f][bad[:3]])
print("density-only pass differs:", int((gd["density"].view(np.uint32) != wd["density"].view(np.uint32)).sum()))
print("input pos min/max", a["pos"].min(0), a["pos"].max(0), "walls", params.xmin, params.xmax)
