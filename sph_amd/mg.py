"""ctypes binding of libsph_mg.so (include/sph_mg.h): the C++ multi-GPU slab driver -- one host thread, one slab per
GPU, RCCL send/recv between x-neighbours.  Replaces MPI_main / MPI_performance_test (SPH/MPI_sph.cpp:245-450).
No CPU fallback: the library needs libsphhip.so and a GPU for every compute call."""
import ctypes
import os

import numpy as np

from . import lib as _lib
from .params import PARTICLE_DTYPE, SphParams

_HERE = os.path.dirname(os.path.abspath(__file__))
MG_PATH = os.path.join(_HERE, "libsph_mg.so")

_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cf = ctypes.c_float
_pi = ctypes.POINTER(ctypes.c_int)

SIGNATURES = {
    "sphmg_create": [ctypes.POINTER(SphParams), _vp, _ci, _ci, _pi, ctypes.POINTER(_vp)],
    "sphmg_step": [_vp, _cf, _cf, _cf, _ci, _ci],
    "sphmg_rebalance": [_vp],
    "sphmg_synchronize": [_vp],
    "sphmg_download": [_vp, _vp, _ci],
    "sphmg_counts": [_vp, _pi],
    "sphmg_bounds": [_vp, _pi],
    "sphmg_rebalance_count": [_vp],
    "sphmg_destroy": [_vp],
    "sphmg_selftest_transport": [_ci, _ci],
    "sphmg_halo_stats": [_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), _pi],
}

_mg = None


def load():
    """Loads libsph_mg.so after libsphhip.so (and torch: one HIP / RCCL runtime per process, see lib.load)."""
    global _mg
    if _mg is not None:
        return _mg
    _lib.load()
    if not os.path.exists(MG_PATH):
        raise _lib.SphHipError("sph_amd/libsph_mg.so is missing: build it with `make -C sph_amd/csrc`")
    mg = ctypes.CDLL(MG_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(mg, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    mg.sphmg_last_error.restype = ctypes.c_char_p
    mg.sphmg_last_error.argtypes = []
    mg.sphmg_transport.restype = ctypes.c_char_p
    mg.sphmg_transport.argtypes = [_vp]
    _mg = mg
    return mg


def _check(rc):
    if rc != 0:
        msg = load().sphmg_last_error()
        raise _lib.SphHipError("sphmg error %d: %s" % (rc, msg.decode() if msg else "?"))


def selftest_transport(device=0, nbytes=1 << 20):
    _check(load().sphmg_selftest_transport(device, nbytes))


class MultiGpuSolver:
    """All slabs of one node driven from this process (the C++ driver); `devices` may name one GPU several times
    (slabs then exchange by device-to-device copies instead of RCCL)."""

    def __init__(self, params, particles, n_gpus=None, devices=None):
        self._mg = load()
        self._h = _vp()
        self.params = params
        self.n = len(particles)
        devices = list(devices) if devices is not None else list(range(n_gpus or 1))
        self.world = len(devices)
        a = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        dev = (_ci * self.world)(*devices)
        _check(self._mg.sphmg_create(ctypes.byref(params), a.ctypes.data, self.n, self.world, dev, ctypes.byref(self._h)))

    def step(self, nsteps=1, mouse=(0.0, 0.0), strength=0.0, rebalance_every=0):
        _check(self._mg.sphmg_step(self._h, mouse[0], mouse[1], strength, nsteps, rebalance_every))

    def rebalance(self):
        _check(self._mg.sphmg_rebalance(self._h))

    def synchronize(self):
        _check(self._mg.sphmg_synchronize(self._h))

    def download(self):
        out = np.zeros(self.n, PARTICLE_DTYPE)
        _check(self._mg.sphmg_download(self._h, out.ctypes.data, self.n))
        return out

    def counts(self):
        c = (_ci * self.world)()
        _check(self._mg.sphmg_counts(self._h, c))
        return list(c)

    def bounds(self):
        b = (_ci * (self.world + 1))()
        _check(self._mg.sphmg_bounds(self._h, b))
        return list(b)

    def halo_stats(self):
        """-> dict: exchanges so far, average bytes one slab sent per exchange, largest steady-state message (records)."""
        ex, by, mx = ctypes.c_longlong(), ctypes.c_double(), _ci()
        _check(self._mg.sphmg_halo_stats(self._h, ctypes.byref(ex), ctypes.byref(by), ctypes.byref(mx)))
        return {"exchanges": ex.value, "bytes_per_exchange_per_slab": by.value, "max_message_records": mx.value}

    @property
    def transport(self):
        return self._mg.sphmg_transport(self._h).decode()

    @property
    def rebalances(self):
        return self._mg.sphmg_rebalance_count(self._h)

    def close(self):
        if self._h:
            self._mg.sphmg_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
