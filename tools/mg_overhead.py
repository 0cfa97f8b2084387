"""Host cost of driving slab steps with neighbours present (C++ driver, libsph_mg.so): S slabs on the one GPU of the
test box, messages by device-to-device copy.  Prints, per slab count: host enqueue time per step (sphmg_step returns
before the GPU is done), wall time per step, and the single-GPU step for the same scene.  The GPU is shared, so wall
time ~ the sum of the slabs' kernels; what matters for a real node is enqueue time per slab per step against the
GPU time of one slab step (one host thread drives all slabs).
    python tools/mg_overhead.py [cols=1024] [steps=200]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.mg import MultiGpuSolver  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p, a = scenes.dam_break(cols)
with Solver(p, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=20)
    s.synchronize()
    t0 = time.perf_counter()
    s.stepSimulation(nsteps=steps)
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
    print("N=%d single GPU: host enqueue %.1f us/step, wall %.1f us/step" % (len(a), (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6), flush=True)
for world in (1, 2, 4, 8):
    with MultiGpuSolver(p, a, devices=[0] * world) as m:
        m.step(20)
        m.synchronize()
        t0 = time.perf_counter()
        m.step(steps)
        t1 = time.perf_counter()
        m.synchronize()
        t2 = time.perf_counter()
        enq, wall = (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6
        print("slabs=%d on one GPU: host enqueue %.1f us/step (%.1f us per slab), wall %.1f us/step, counts %s" % (world, enq, enq / world, wall, m.counts()), flush=True)
