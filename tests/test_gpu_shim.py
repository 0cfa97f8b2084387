"""The C++ drop-in layer (include/sph_hip.h, sph_amd/csrc/host_shim.cpp) driven through a small test
executable, checked against the golden vectors / oracle; and the headless dispatcher."""
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from sph_amd.params import PARTICLE_DTYPE

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "shim_driver")
CLI = os.path.join(ROOT, "sph_amd", "sph_main")
O = orc.Oracle()
P = O.default_params()


def G(name):
    return np.load(os.path.join(orc.GOLDEN, name + ".npz"), allow_pickle=False)


def drive(tmp_path, op, a, n, *args):
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    if a is not None:
        np.ascontiguousarray(a).tofile(fin)
    subprocess.run([DRIVER, op, fin if a is not None else "-", fout, str(n)] + [str(x) for x in args], check=True,
                   timeout=120)
    return fout


def test_init_matches_reference_golden(tmp_path):
    g = G("f1_init")
    for k in ("n1", "n17", "n1000", "n1024", "n2704"):
        n = int(k[1:])
        out = np.fromfile(drive(tmp_path, "init", None, n), PARTICLE_DTYPE)
        assert orc.same_bits(out, g[k]), k


def test_phases_through_shim(tmp_path):
    g = G("f4_phases_n1024")
    n = 1024
    d = np.fromfile(drive(tmp_path, "density", g["state"], n), PARTICLE_DTYPE)
    want, _ = O.density_cells(P, g["state"].copy())
    assert orc.same_bits(d, want)
    f = np.fromfile(drive(tmp_path, "forces", g["after_density"], n, 1, 1.0, 1.0, 50000.0), PARTICLE_DTYPE)
    assert orc.same_bits(f, O.forces_cells(P, g["after_density"].copy(), (1.0, 1.0), 50000.0))
    i = np.fromfile(drive(tmp_path, "integrate", g["after_forces"], n), PARTICLE_DTYPE)
    assert orc.same_bits(i, g["after_integrate"])  # integrate is order-free: equals the reference bit for bit


def test_step_through_shim(tmp_path):
    a = G("f2_serial_traj_n1024")["s0"]
    want = O.step_cells(P, a.copy(), 5)
    one_by_one = np.fromfile(drive(tmp_path, "step", a, 1024, 5), PARTICLE_DTYPE)
    resident = np.fromfile(drive(tmp_path, "resident", a, 1024, 5), PARTICLE_DTYPE)
    assert orc.same_bits(one_by_one, want) and orc.same_bits(resident, want)


def test_cells_through_shim(tmp_path):
    a = G("f2_serial_traj_n1024")["s100"]
    raw = np.fromfile(drive(tmp_path, "cells", a, 1024), np.int32)
    nc = raw[0]
    keys, order, cs, ce = np.split(raw[1:], [1024, 2048, 2048 + nc])
    okeys, oorder, oskeys, ocs = O.build_cells(P, a)
    assert nc == len(ocs) - 1
    assert np.array_equal(keys, oskeys) and np.array_equal(order, oorder)
    assert np.array_equal(cs, ocs[:-1]) and np.array_equal(ce, ocs[1:])


def test_device_flavour_interface(tmp_path):
    """initSimulation / stepSimulation shapes of the CUDA variant (SPH/sph.cuh:37-39): resident state, host
    array refreshed by every call, vbo contents of updateVBO, and the (in the reference dead) grid /
    cellStart / cellEnd parameters describing the returned state."""
    n, steps = 1024, 4
    fout = drive(tmp_path, "device", None, n, steps, 1.0, 1.0, 50000.0)
    raw = open(fout, "rb").read()
    got = np.frombuffer(raw[:28 * n], PARTICLE_DTYPE)
    rest = np.frombuffer(raw[28 * n:], np.int32)
    grid, nc = rest[:n], int(rest[n])
    cs, ce = rest[n + 1:n + 1 + nc], rest[n + 1 + nc:n + 1 + 2 * nc]
    vbo = rest[n + 1 + 2 * nc:].view(np.float32).reshape(n, 2)
    a = G("f1_init")["n1024"]
    want = a.copy()
    for _ in range(steps):
        want = O.step_cells(P, want, 1, mouse=(1.0, 1.0), strength=50000.0)
    assert orc.same_bits(got, want)
    okeys, oorder, oskeys, ocs = O.build_cells(P, want)
    assert nc == len(ocs) - 1
    assert np.array_equal(grid, okeys)
    assert np.array_equal(cs, ocs[:-1]) and np.array_equal(ce, ocs[1:])
    assert np.array_equal(vbo.view(np.uint32), got["pos"].view(np.uint32))


def test_headless_dispatcher(tmp_path):
    out = subprocess.run([CLI, "--n", "1024", "--mode", "perf"], capture_output=True, text=True, check=True,
                         timeout=300).stdout
    m = re.search(r"HIP Average UPS: ([0-9.e+]+)", out)  # "X Average UPS", SPH/main.cpp:115
    assert m and float(m.group(1)) > 0
    assert out.count("Run: ") == 10  # 10 runs, serial_main.cpp:125
    dump = str(tmp_path / "state.bin")
    subprocess.run([CLI, "--n", "1024", "--mode", "run", "--steps", "10", "--dump", dump], check=True, timeout=120)
    got = np.fromfile(dump, PARTICLE_DTYPE)
    assert orc.same_bits(got, O.step_cells(P, O.init_lattice(1024), 10))


def test_scripted_mouse_and_frames(tmp_path):
    """Scripted stand-in for the interactive loop (serial_main.cpp:68-105): mouse events by step, pixel ->
    simulation mapping of serial_main.cpp:72-75, frames as PPM files.  The final state must equal the
    oracle stepped with the same per-step mouse input, bit for bit."""
    script = [(0, 400.0, 400.0, 0), (5, 300.0, 500.0, 1), (12, 520.0, 260.0, 2), (20, 0.0, 0.0, 0)]
    sp = tmp_path / "mouse.txt"
    sp.write_text("".join("%d %g %g %d\n" % e for e in script))
    dump, prefix = str(tmp_path / "state.bin"), str(tmp_path / "frame_")
    steps = 26
    subprocess.run([CLI, "--n", "1024", "--mode", "run", "--steps", str(steps), "--mouse-script", str(sp), "--frames", "8",
                    "--frame-prefix", prefix, "--dump", dump], check=True, timeout=120)
    a = O.init_lattice(1024)
    for s in range(steps):
        _, px, py, button = [e for e in script if e[0] <= s][-1]
        mx = np.float32((px / 800.0) * 2.0)
        my = np.float32(2.0 - (py / 800.0) * 2.0)
        strength = {0: 0.0, 1: 50000.0, 2: -50000.0}[button]
        a = O.step_cells(P, a, 1, (float(mx), float(my)), strength)
    got = np.fromfile(dump, PARTICLE_DTYPE)
    assert orc.same_bits(got, a)
    for s in (8, 16, 24, 26):  # every 8 steps and after the last one
        raw = open(prefix + "%06d.ppm" % s, "rb").read()
        assert raw.startswith(b"P6\n800 800\n255\n") and len(raw) == 15 + 800 * 800 * 3
        img = np.frombuffer(raw[15:], np.uint8).reshape(800, 800, 3)
        lit = (img[:, :, 2] == 255).sum()
        assert 900 <= lit <= 1024  # one pixel per particle (a few may share a pixel)
