"""Workload for a kernel trace of the slab path: S slabs on one GPU (C++ driver, copy transport), K steps.
    rocprofv3 --kernel-trace --stats -- python3 tools/mg_trace.py [cols=1024] [slabs=2] [steps=100]"""
import sys

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.mg import MultiGpuSolver  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
p, a = scenes.dam_break(cols)
with MultiGpuSolver(p, a, devices=[0] * world) as m:
    m.step(steps)
    m.synchronize()
    print("done", m.counts())
