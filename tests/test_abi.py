"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/sphhip.h declares; the host-side derivation reproduces the reference's folded
constants; argument validation works without touching a device."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from sph_amd import lib, scenes
from sph_amd.params import PARTICLE_DTYPE, SphDerived, SphParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sphhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sphhip_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "libsphhip.so does not export %s" % s
    # and the Python binding table covers the header (sphhip_last_error is bound separately)
    assert set(syms) - {"sphhip_last_error"} == set(lib.SIGNATURES)


def test_struct_sizes_match_header():
    assert ctypes.sizeof(SphParams) == 18 * 4
    assert ctypes.sizeof(SphDerived) == 11 * 4 + 2 * 4


def test_default_params_are_the_reference_defines():
    p = lib.default_params()
    r = SphParams.reference()
    assert bytes(p) == bytes(r)
    # SPH/sph.h:11-20
    assert (p.h, p.dt, p.mass) == (np.float32(0.15), np.float32(0.0005), np.float32(0.05))
    assert (p.stiffness, p.rest_density, p.max_vel) == (1000.0, 1000.0, 5.0)


def test_derived_constants_bits():
    """SURVEY.md section 7 'Coefficients via powf': values at H=0.15f."""
    d = lib.derive(lib.default_params())
    hx = lambda v: np.float32(v).view(np.uint32)  # noqa: E731
    assert hx(d.h2) == 0x3CB851EC
    assert hx(d.poly6) == 0x4C1B75DA
    assert hx(d.spiky) == 0xC999818A
    assert hx(d.visc_lap) == 0x4999818A
    assert hx(d.visc_mass) == 0x3C75C290
    assert hx(d.lj_cut) == 0x3E000000
    assert (d.nx, d.ny) == (13, 13)
    assert d.cell > 0.15


def test_derive_rejects_bad_params():
    p = lib.default_params()
    p.cell_size = 0.1  # < h: the 3x3 stencil would miss in-range pairs
    with pytest.raises(lib.SphHipError):
        lib.derive(p)
    p = lib.default_params()
    p.xmax = p.xmin
    with pytest.raises(lib.SphHipError):
        lib.derive(p)


def test_null_arguments_are_rejected_without_a_gpu():
    L = lib.load()
    assert L.sphhip_create(None, 10, 0, None) == -1
    assert L.sphhip_upload(None, None, 0, 0) == -1
    assert L.sphhip_step(None, 0.0, 0.0, 0.0, 1) == -1
    assert b"null" in L.sphhip_last_error()


def test_scene_generators():
    a = scenes.lattice(1024)
    assert a.dtype == PARTICLE_DTYPE and len(a) == 1024
    p, a = scenes.dam_break(32)
    d = lib.derive(p)
    assert len(a) == 1024 and a["pos"][:, 0].max() < p.xmax and a["pos"][:, 1].max() < p.ymax
    sp = np.float32(0.15) / np.float32(4)
    assert np.isclose(a["pos"][1, 0] - a["pos"][0, 0], sp)
    assert d.nx > 2 * d.ny - 3  # 4W x 2W box
    p, a = scenes.uniform_box(40)
    assert len(a) == 1600 and np.isclose(a["pos"][:, 0].max(), p.xmax, atol=1e-4)


def test_host_shim_exports_reference_shaped_functions():
    so = os.path.join(ROOT, "sph_amd", "libsph_hip_host.so")
    if not os.path.exists(so):
        pytest.skip("host shim not built")
    out = subprocess.run(["nm", "-D", "--demangle", so], capture_output=True, text=True, check=True).stdout
    for name in ("hip_initSimulation(", "hip_buildNeighbors(", "hip_computeDensityPressure(", "hip_computeForces(",
                 "hip_integrate(", "hip_stepSimulation(", "hip_cleanupSimulation(", "HIP_main(int)",
                 "HIP_performance_test(int)"):
        assert name in out, name


def test_mg_library_exports_every_declared_symbol():
    """libsph_mg.so (the C++ multi-GPU slab driver) loads without a GPU and exports what include/sph_mg.h declares."""
    from sph_amd import mg

    M = mg.load()
    src = open(os.path.join(ROOT, "include", "sph_mg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(sphmg_[a-z_]+)\s*\(", src)))
    assert len(syms) == 13
    for s in syms:
        assert hasattr(M, s), "libsph_mg.so does not export %s" % s
    assert set(syms) - {"sphmg_last_error", "sphmg_transport"} == set(mg.SIGNATURES)
    assert M.sphmg_transport(None) == b"none"
    # argument validation needs no device
    assert M.sphmg_step(None, 0.0, 0.0, 0.0, 1, 0) != 0 and b"bad argument" in M.sphmg_last_error()
    assert M.sphmg_rebalance_count(None) == 0
