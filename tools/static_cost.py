"""What a call of sphhip_step_static costs beside the same step on a context (the entry point that allocates nothing
rebuilds its context and clears the used part of the static scratch array in every call).  python tools/static_cost.py"""
import ctypes
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sph_amd import lib as _lib, scenes  # noqa: E402
from sph_amd.params import SphParams  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
for label, (params, a) in (("lattice 4096 (reference box)", (SphParams.reference(), scenes.lattice(4096))),
                           ("dam break 65536 (the stated maximum)", scenes.dam_break(256))):
    raw = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    out = torch.empty_like(raw)
    torch.cuda.synchronize()

    def static(nsteps, calls):
        t0 = time.perf_counter()
        for _ in range(calls):
            _lib.check(L.sphhip_step_static(ctypes.byref(params), 0, ctypes.c_void_p(raw.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            len(a), 0.0, 0.0, 0.0, nsteps, None))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / calls * 1e3

    static(1, 3)
    s1, s8 = static(1, 200), static(8, 50)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=3)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            s.stepSimulation(nsteps=1)
        s.synchronize()
        c1 = (time.perf_counter() - t0) / 200 * 1e3
    print("%s: static call of 1 step %.3f ms, of 8 steps %.3f ms (%.3f per step); context step (resident state, no unpack / pack) %.3f ms"
          % (label, s1, s8, s8 / 8, c1), flush=True)
