"""wall time per step of S slabs on one GPU (C++ driver): python tools/mg_quick.py cols slabs steps"""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa
from sph_amd import scenes
from sph_amd.mg import MultiGpuSolver
cols, world, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p, a = scenes.dam_break(cols)
with MultiGpuSolver(p, a, devices=[0] * world) as m:
    m.step(20); m.synchronize()
    t0 = time.perf_counter(); m.step(steps); t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
    print("slabs=%d: enqueue %.1f us/step, wall %.1f us/step" % (world, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6), flush=True)
