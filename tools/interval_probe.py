"""Where inside bench.py's timed intervals the time goes: the 1M dam break after the same set-up (prime, re-upload,
warmup), timed in blocks of `block` steps with the tiling the library chose in each.  python tools/interval_probe.py [block]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

block = int(sys.argv[1]) if len(sys.argv) > 1 else 10
params, a = scenes.dam_break(1024, 1024)
with Solver(params, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=150)
    s.upload(a)
    s.stepSimulation(nsteps=5)
    s.synchronize()
    done = 5
    for _ in range(300 // block):
        t0 = time.perf_counter()
        s.stepSimulation(nsteps=block)
        s.synchronize()
        dt = (time.perf_counter() - t0) / block
        done += block
        st = s.tile_stats()
        print("steps %3d..%3d: %.4f ms/step  tiles %d rows %d long %d dense %d/%d" % (
            done - block + 1, done, dt * 1e3, st["tiles"], st["tile_rows"], st["long"], st["dense_density"], st["dense_force"]), flush=True)

# the same in bench.py's shape: 50-step intervals with the force kernel bracketed by events in one step of eight;
# "flush": the phase times are read (events recycled) between the intervals, and the events exist before the first one
for mode in ("bench", "flush"):
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=150)
        s.upload(a)
        s.set_profiling_phases(["force"])
        s.set_profiling_period(8)
        if mode == "flush":
            s.set_profiling_period(1)  # (5 event pairs come into being here)
            s.reset_phase_times()
            s.stepSimulation(nsteps=5)
            s.synchronize()
            s.set_profiling_period(8)
            s.reset_phase_times()
        else:
            s.set_profiling(False)
            s.stepSimulation(nsteps=5)
            s.synchronize()
            s.set_profiling_phases(["force"])
            s.set_profiling_period(8)
            s.reset_phase_times()
        out = []
        for _ in range(6):
            t0 = time.perf_counter()
            s.stepSimulation(nsteps=50)
            s.synchronize()
            out.append((time.perf_counter() - t0) / 50 * 1e3)
            if mode == "flush":
                s.phase_times()
        print(mode, " ".join("%.4f" % x for x in out), "force", s.phase_times()["force"], flush=True)
