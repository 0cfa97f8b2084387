import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sph_amd import scenes
from sph_amd.solver import Solver
cols, steps = int(sys.argv[1]), int(sys.argv[2])
params, a = scenes.dam_break(cols)
with Solver(params, n_max=len(a)) as s:
    s.upload(a); s.stepSimulation(nsteps=steps)
    s.set_profiling(True); s.reset_phase_times() if hasattr(s, "reset_phase_times") else None
    s.stepSimulation(nsteps=20); s.synchronize()
    print({k: round(v[0] / max(v[1], 1), 4) for k, v in s.phase_times().items()})
    np.save("/root/repo/gpurun_out/state_%d_%d.npy" % (cols, steps), s.download())
