"""Evolved 1M dam break: ms/step and work-list statistics after E steps:
    python tools/evolved_quick.py [E=1000] [K=60] [chunk=10] [profile=0]
K timed steps in calls of `chunk` steps (chunk = K: one call, as bench.py does); profile=1: HIP events around the force pass."""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa
from sph_amd import scenes
from sph_amd.solver import Solver
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prof = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p, a = scenes.dam_break(1024)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=E)
    s.stepSimulation(nsteps=5)
    s.synchronize()
    if prof:
        s.set_profiling_phases(["force"])
        s.reset_phase_times()
    seen = []
    t0 = time.perf_counter()
    for i in range(K // chunk):
        s.stepSimulation(nsteps=chunk)
        if chunk < K:
            seen.append(s.tile_stats())
    s.synchronize()
    dt = (time.perf_counter() - t0) / K
    seen.append(s.tile_stats())
    print("after %d steps, calls of %d, profile %d: %.4f ms/step; long %s dense_density %s dense_force %s%s" % (
        E, chunk, prof, dt * 1e3, [x["long"] for x in seen], [x["dense_density"] for x in seen], [x["dense_force"] for x in seen],
        ("; force %.4f ms" % (s.phase_times()["force"][0] / K)) if prof else ""), flush=True)
