"""The C++ multi-GPU slab driver (libsph_mg.so, include/sph_mg.h) on the one GPU of the test box: with one slab it is
the single-GPU path; with several slabs on the same device it runs the whole protocol (balanced boundaries, split
step, packed halo + migrant messages, append, rebalance, gather) with device-to-device copies as the transport; the
RCCL transport itself is checked by a loopback send/recv.  Results must equal the single-GPU run and the oracle bit
for bit.  (RCCL between distinct GPUs has not run on hardware: the test box has one.)"""
import os
import subprocess

import numpy as np
import pytest

import orc
from sph_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene():
    p, a = scenes.dam_break(96, 64)
    a["vel"][:, 0] = 4.0
    a["pos"][:, 0] += np.float32(3.0)
    a["valid"][::11] = 0
    return p, a


def _single(params, a, nsteps):
    from sph_amd.solver import Solver

    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=nsteps)
        return s.download()


def test_transport_loopback():
    from sph_amd import mg

    mg.selftest_transport(0, 1 << 20)
    mg.selftest_transport(0, 20 * 4097)


def test_one_slab_is_the_single_gpu_path():
    from sph_amd.mg import MultiGpuSolver

    params, a = _scene()
    with MultiGpuSolver(params, a, devices=[0]) as m:
        m.step(20)
        m.synchronize()
        got = m.download()
        assert m.counts() == [len(a)]
    assert orc.same_bits(got, _single(params, a, 20))
    assert orc.same_bits(got, orc.Oracle().step_cells(params, a.copy(), 20))


@pytest.mark.parametrize("world,nsteps,every", [(2, 30, 0), (3, 30, 10), (4, 90, 30)])
def test_slabs_on_one_device_equal_single_gpu(world, nsteps, every):
    from sph_amd.mg import MultiGpuSolver

    params, a = _scene()
    with MultiGpuSolver(params, a, devices=[0] * world) as m:
        b0 = m.bounds()
        c0 = m.counts()
        assert sum(c0) == len(a) and b0[0] == 0 and all(b0[i + 1] - b0[i] >= 3 for i in range(world))
        m.step(nsteps // 3)
        mid = m.download()  # a gather between two runs must not disturb the state
        m.step(nsteps - nsteps // 3, rebalance_every=every)
        m.synchronize()
        got = m.download()
        assert sum(m.counts()) == len(a)
        if every:
            assert m.rebalances >= 1
    assert orc.same_bits(mid, _single(params, a, nsteps // 3))
    single = _single(params, a, nsteps)
    assert orc.same_bits(got, single)
    if nsteps <= 30:
        assert orc.same_bits(got, orc.Oracle().step_cells(params, a.copy(), nsteps))


def test_too_many_slabs_are_refused():
    from sph_amd import lib
    from sph_amd.mg import MultiGpuSolver
    from sph_amd.params import SphParams

    a = scenes.lattice(1024)  # the reference box has 13 columns: at most 4 slabs
    with pytest.raises(lib.SphHipError):
        MultiGpuSolver(SphParams.reference(), a, devices=[0] * 5)


def test_sph_main_methods(tmp_path):
    exe = os.path.join(ROOT, "sph_amd", "sph_main")
    out = subprocess.run([exe, "--method", "hip-mg", "--gpus", "1", "--n", "1024", "--mode", "perf"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "HIP-MG Average UPS:" in out.stdout
    d1, d2 = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    for method, d in (("hip", d1), ("hip-mg", d2)):
        r = subprocess.run([exe, "--method", method, "--n", "1024", "--mode", "run", "--steps", "50", "--dump", d], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    assert open(d1, "rb").read() == open(d2, "rb").read()
    bad = subprocess.run([exe, "--method", "cuda"], capture_output=True, text=True)
    assert bad.returncode == 2


def test_transport_selection(monkeypatch):
    """SPHMG_TRANSPORT: RCCL needs one device per slab (refused when slabs share a device); `copy` is accepted; anything
    else is an error."""
    from sph_amd import lib
    from sph_amd.mg import MultiGpuSolver

    params, a = _scene()
    monkeypatch.setenv("SPHMG_TRANSPORT", "rccl")
    with pytest.raises(lib.SphHipError):
        MultiGpuSolver(params, a, devices=[0, 0])
    monkeypatch.setenv("SPHMG_TRANSPORT", "bogus")
    with pytest.raises(lib.SphHipError):
        MultiGpuSolver(params, a, devices=[0, 0])
    monkeypatch.setenv("SPHMG_TRANSPORT", "copy")
    with MultiGpuSolver(params, a, devices=[0, 0]) as m:
        m.step(3)
        got = m.download()
    assert orc.same_bits(got, _single(params, a, 3))


def test_dam_scene_of_the_dispatcher_equals_the_python_scene(tmp_path):
    """sph_main --scene dam builds the dam-break lattice in C++ (sph_main.cpp): same bits as sph_amd.scenes.dam_break,
    checked through a 2-step run of both methods against the oracle."""
    exe = os.path.join(ROOT, "sph_amd", "sph_main")
    params, a = scenes.dam_break(64)
    want = orc.Oracle().step_cells(params, a.copy(), 2)
    for method in ("hip", "hip-mg"):
        dump = str(tmp_path / (method + ".bin"))
        r = subprocess.run([exe, "--method", method, "--scene", "dam", "--n", "4096", "--mode", "run", "--steps", "2", "--dump", dump],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(dump, dtype=a.dtype)
        assert orc.same_bits(got, want), method
