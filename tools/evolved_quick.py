"""Evolved 1M dam break: ms/step and work-list statistics after E steps: python tools/evolved_quick.py [E=1000] [K=60]"""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa
from sph_amd import scenes
from sph_amd.solver import Solver
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
p, a = scenes.dam_break(1024)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=E)
    s.synchronize()
    seen = []
    t0 = time.perf_counter()
    for i in range(K // 10):
        s.stepSimulation(nsteps=10)
        seen.append(s.tile_stats())
    s.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("after %d steps: %.4f ms/step; long %s dense_density %s dense_force %s" % (
        E, dt * 1e3, [x["long"] for x in seen], [x["dense_density"] for x in seen], [x["dense_force"] for x in seen]), flush=True)
