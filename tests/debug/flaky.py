"""Test infrastructure: runs oracle and HIP path twice each per scaled fuzz scene and reports which side is
not reproducible:  python tests/debug/flaky.py <first> <last + 1> <repeats>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orc
from sph_amd import scenes
from sph_amd.solver import Solver
O = orc.Oracle()
lo, hi, rep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cnt = {"oracle": 0, "gpu": 0, "cross": 0}
for r in range(rep):
    for seed in range(lo, hi):
        rng = np.random.default_rng(5000 + seed)
        cols = int(rng.choice([8, 20, 33, 64, 100, 150]))
        params, a = scenes.dam_break(cols, int(rng.integers(4, cols + 1))) if seed % 2 else scenes.uniform_box(cols)
        a["pos"] += rng.uniform(-0.01, 0.01, size=a["pos"].shape).astype(np.float32)
        a["vel"] = rng.normal(0.0, float(rng.choice([0.0, 0.5, 3.0])), size=a["vel"].shape).astype(np.float32)
        steps = int(rng.choice([1, 7, 18, 27, 40]))
        w1 = O.step_cells(params, a.copy(), steps); w2 = O.step_cells(params, a.copy(), steps)
        g = []
        for _ in range(2):
            with Solver(params, n_max=len(a)) as s:
                s.upload(a); s.stepSimulation(nsteps=steps); g.append(s.download())
        if not orc.same_bits_nan(w1, w2): cnt["oracle"] += 1; print("ORACLE differs", r, seed, flush=True)
        if not orc.same_bits_nan(g[0], g[1]): cnt["gpu"] += 1; print("GPU differs", r, seed, len(a), steps, flush=True)
        if not orc.same_bits_nan(g[0], w1): cnt["cross"] += 1; print("GPU vs oracle differs", r, seed, flush=True)
print("done", cnt)
