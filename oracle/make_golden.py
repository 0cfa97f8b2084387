#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's own serial step (oracle/_ref/libsph_ref.so,
i.e. /root/reference/SPH/sph.cpp compiled unmodified by oracle/Makefile).  TEST INFRASTRUCTURE.

Run here (the container that has /root/reference):   make -C oracle && python oracle/make_golden.py
The fixtures are data only (inputs + outputs of the reference); no reference source travels.

The reference has no snapshot-semantics (MPI) build here -- SPH/MPI_sph.cpp needs mpi.h and GL
libraries the image lacks, so it is unbuildable -- and no neighbour-count function at all, so the
two fixtures marked `by="oracle"` are produced by oracle/sph_oracle.c (which is itself pinned bit
for bit against the reference build by tests/test_oracle_pins.py) and say so in their metadata.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from sph_amd.params import PARTICLE_DTYPE  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
R = orc.Reference()
O = orc.Oracle()
P = O.default_params()


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in kw.items()})


def block(nx, ny, spacing, x0, y0, rho=1000.0):
    a = np.zeros(nx * ny, PARTICLE_DTYPE)
    i = np.arange(nx * ny)
    a["pos"][:, 0] = (np.float32(x0) + (i % nx).astype(np.float32) * np.float32(spacing)).astype(np.float32)
    a["pos"][:, 1] = (np.float32(y0) + (i // nx).astype(np.float32) * np.float32(spacing)).astype(np.float32)
    a["density"] = rho
    a["valid"] = 1
    return a


# F1: lattice init (serial_initSimulation, sph.cpp:132-164)
f1 = {}
for n in (1, 2, 3, 5, 17, 1000, 1024, 2704):
    f1["n%d" % n] = R.init(n)
save("f1_init", **f1)

# F2: serial trajectory, default scene N=1024 (main.cpp:68)
a = R.init(1024)
f2 = {"s0": a.copy()}
done = 0
for s in (1, 10, 50, 100, 200):
    R.step(a, s - done)
    done = s
    f2["s%d" % s] = a.copy()
save("f2_serial_traj_n1024", **f2)

# F3: snapshot-semantics trajectory (mpi_computeStep, MPI_sph.cpp:135-243) -- by the oracle; the
# survey's recorded 1-rank MPI values after 50 steps are kept beside it as an external pin.
a = O.init_lattice(1024, sqrt_f32=True)
f3 = {"s0": a.copy()}
done = 0
for s in (1, 10, 50):
    O.step_allpairs(P, a, s - done, flags=orc.SNAPSHOT | orc.EPS_F32)
    done = s
    f3["s%d" % s] = a.copy()
f3["survey_pin_s50_p0v0"] = np.array([0.0502954237, 1.94651711, 0.0230582077, -0.273058176], np.float32)
f3["by"] = np.array("oracle")
save("f3_snapshot_traj_n1024", **f3)

# F4: single phases from a mid-run state (after 50 serial steps)
mid = f2["s50"].copy()
d = R.density(mid.copy())
f = R.forces(d.copy())
g = R.integrate(f.copy())
save("f4_phases_n1024", state=mid, after_density=d, after_forces=f, after_integrate=g)

# F5: dense block (pressure > 0; spacing 0.012 => ~490 neighbours) and a moderately dense one
dense = block(40, 40, 0.012, 0.6, 0.6)
rng = np.random.default_rng(1234)
dense["vel"] = rng.uniform(-0.5, 0.5, (len(dense), 2)).astype(np.float32)
d = R.density(dense.copy())
f = R.forces(d.copy())
g = R.integrate(f.copy())
s3 = R.step(dense.copy(), 3)
save("f5_dense_block", state=dense, after_density=d, after_forces=f, after_integrate=g, after_3steps=s3)

# F6: mouse interaction (serial_main.cpp:72-75: strength +-50000)
st = f2["s10"].copy()
d = R.density(st.copy())
push = R.forces(d.copy(), (1.0, 1.0), 50000.0)
pull = R.forces(d.copy(), (0.7, 1.3), -50000.0)
save("f6_mouse", state=d, push=push, pull=pull, push_args=np.array([1.0, 1.0, 50000.0], np.float32),
     pull_args=np.array([0.7, 1.3, -50000.0], np.float32))

# F7: walls (all four, corners) and the MAX_VEL cap (serial_integrate, sph.cpp:101-130)
w = np.zeros(12, PARTICLE_DTYPE)
w["valid"] = 1
w["density"] = 1000.0
w["pos"] = np.array([[0.05, 1.0], [1.95, 1.0], [1.0, 0.05], [1.0, 1.95], [0.0501, 0.0501], [1.9499, 1.9499],
                     [1.0, 1.0], [1.0, 1.0], [0.04, 2.0], [0.05, 1.95], [1.0, 1.0], [1.94999, 1.0]], np.float32)
w["vel"] = np.array([[-3.0, 0.5], [4.0, -0.5], [0.5, -2.0], [0.25, 4.5], [-4.0, -4.0], [4.0, 4.0],
                     [30.0, 40.0], [-6.0, 0.0], [-1.0, 1.0], [0.0, 0.0], [3.0, 4.0], [100.0, 0.0]], np.float32)
w["valid"][10] = 0
save("f7_walls", state=w, after_integrate=R.integrate(w.copy()))

# F8: neighbour counts #{j: r2 < H*H} incl. self -- reference predicate (sph.cpp:46-48) evaluated
# by the oracle's all-pairs loop (the reference has no function that returns counts).
save("f8_neighbor_counts", by=np.array("oracle"),
     n1024_s1=O.neighbor_counts_allpairs(P, f2["s1"]), n1024_s100=O.neighbor_counts_allpairs(P, f2["s100"]),
     dense=O.neighbor_counts_allpairs(P, dense))

# F9: some valid=false particles (sph.cpp:59,63,103)
v = f2["s10"].copy()
v["valid"][::7] = 0
save("f9_invalid", state=v, after_1step=R.step(v.copy(), 1), after_5steps=R.step(v.copy(), 5))

# F10: pair-kernel tables (sph.cpp:7-39) at H, SIGMA, EPSILON of sph.h
h = np.float32(0.15)
r = np.unique(np.concatenate([
    np.array([0.0, 1e-7, 9.99999997e-7, 1.0000001e-6, 1e-5, 0.05, 0.1249999, 0.125, 0.1250001, 0.1499999, 0.15,
              0.1500001, 0.2], np.float32),
    np.linspace(0.001, 0.16, 51, dtype=np.float32)]))
poly6 = np.array([R.L.ref_poly6(np.float32(x * x), h) for x in r], np.float32)
visc = np.array([R.L.ref_visc_lap(x, h) for x in r], np.float32)
spiky = np.array([R.spiky_grad(np.float32(0.6) * x, np.float32(0.8) * x, x, h) for x in r], np.float32)
lj = np.array([R.lj(np.float32(0.6) * x, np.float32(0.8) * x, x, np.float32(0.05), np.float32(1.0)) for x in r],
              np.float32)
save("f10_kernel_tables", r=r, poly6=poly6, visc=visc, spiky=spiky, lj=lj)

# F11: other particle counts, short horizons (non-square lattices, higher packing)
f11 = {}
for n, s in ((1000, 3), (2704, 2), (4096, 3)):
    a = R.init(n)
    R.step(a, s)
    f11["n%d_s%d" % (n, s)] = a
save("f11_short_runs", **f11)
