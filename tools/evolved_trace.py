"""Workload for a kernel trace of the evolved scene: the 1M dam break after E steps, then K more steps.
    rocprofv3 --kernel-trace --stats -- python3 tools/evolved_trace.py [cols=1024] [evolve=1000] [steps=20]"""
import sys

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
evolve = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
p, a = scenes.dam_break(cols)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=evolve)
    s.synchronize()
    print("evolved", flush=True)
    s.stepSimulation(nsteps=steps)
    s.synchronize()
