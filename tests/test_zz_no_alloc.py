"""sphhip_step_static -- the entry point that allocates nothing (include/sphhip.h) -- against the existing
entry points it is a variant of (sphhip_upload + sphhip_step + sphhip_download), through the C ABI, bit for bit:
two calls in a row on the same GPU, the second with a smaller input (the static scratch array keeps the first
call's contents), then in place, then the stated limits.  Collected last (test_zz_*): a failure here must not
hide the other tests from `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch  # before libsphhip.so: one HIP runtime per process (sph_amd/lib.py)

import orc
from sph_amd import lib as _lib, scenes
from sph_amd.params import PARTICLE_DTYPE, SphParams
from sph_amd.solver import Solver

pytestmark = pytest.mark.gpu


def _static_step(L, params, a, nsteps, mouse=(0.0, 0.0), strength=0.0, in_place=False):
    dev = torch.device("cuda", 0)
    raw = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    out = raw if in_place else torch.full_like(raw, 0xEE)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev).cuda_stream  # 0: the null stream
    _lib.check(L.sphhip_step_static(ctypes.byref(params), 0, ctypes.c_void_p(raw.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                    len(a), mouse[0], mouse[1], strength, nsteps, ctypes.c_void_p(stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(PARTICLE_DTYPE).copy()


def _context_step(params, a, nsteps, mouse=(0.0, 0.0), strength=0.0):
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(mouse, strength, nsteps)
        return s.download()


def test_static_step_equals_context_step_twice_in_a_row():
    L = _lib.load()
    P = SphParams.reference()
    # first call: a dam break at the stated maximum (65536 particles, scaled walls); second call: SMALLER input,
    # other parameters (the reference box), mouse force, invalid particles -- on top of the first call's leftovers
    params1, a1 = scenes.dam_break(256)
    assert len(a1) == 65536
    got1 = _static_step(L, params1, a1, 5)
    a2 = scenes.lattice(3000)
    a2["valid"][::7] = 0
    got2 = _static_step(L, P, a2, 4, (1.0, 1.0), 50000.0)
    assert orc.same_bits(got1, _context_step(params1, a1, 5))
    assert orc.same_bits(got2, _context_step(P, a2, 4, (1.0, 1.0), 50000.0))
    # ... and against the oracle directly
    O = orc.Oracle()
    assert orc.same_bits(got2, O.step_cells(P, a2.copy(), 4, mouse=(1.0, 1.0), strength=50000.0))
    # in place (input array = output array), a dense scene (dense sub-tile kernels), one particle
    a3 = scenes.lattice(8192)
    assert orc.same_bits(_static_step(L, P, a3, 2, in_place=True), _context_step(P, a3, 2))
    a4 = scenes.lattice(1)
    assert orc.same_bits(_static_step(L, P, a4, 3), _context_step(P, a4, 3))


def test_static_step_limits():
    L = _lib.load()
    P = SphParams.reference()
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ptr = ctypes.c_void_p(buf.data_ptr())
    # beyond the stated maxima: a clear error, nothing enqueued
    assert L.sphhip_step_static(ctypes.byref(P), 0, ptr, ptr, 65537, 0.0, 0.0, 0.0, 1, None) == -4
    assert b"SPHHIP_STATIC_MAX_N" in L.sphhip_last_error()
    big, _ = scenes.dam_break(1024)  # 1021 x 511 cells
    assert L.sphhip_step_static(ctypes.byref(big), 0, ptr, ptr, 2, 0.0, 0.0, 0.0, 1, None) == -4
    assert b"SPHHIP_STATIC_MAX_CELLS" in L.sphhip_last_error()
    assert L.sphhip_step_static(ctypes.byref(P), 0, None, ptr, 2, 0.0, 0.0, 0.0, 1, None) == -1
    assert L.sphhip_step_static(ctypes.byref(P), 0, ptr, ptr, 0, 0.0, 0.0, 0.0, 1, None) == 0  # empty input: nothing to do
