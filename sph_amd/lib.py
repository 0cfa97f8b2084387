"""ctypes binding of libsphhip.so (include/sphhip.h).  There is no CPU fallback: if the HIP
library is missing this module raises at import of the symbol table, and every compute call needs
a GPU."""
import ctypes
import os
import sys

from .params import SphDerived, SphParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPHHIP_LIB") or os.path.join(_HERE, "libsphhip.so")  # override: kernel A/B builds only

PH_NAMES = ["hash", "sort", "cells", "reorder", "density", "force", "integrate", "force_interior"]

_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cf = ctypes.c_float
_pi = ctypes.POINTER(ctypes.c_int)

# name -> argtypes; every function returns int except sphhip_last_error
SIGNATURES = {
    "sphhip_default_params": [ctypes.POINTER(SphParams)],
    "sphhip_derive": [ctypes.POINTER(SphParams), ctypes.POINTER(SphDerived)],
    "sphhip_create": [ctypes.POINTER(SphParams), _ci, _ci, ctypes.POINTER(_vp)],
    "sphhip_destroy": [_vp],
    "sphhip_set_stream": [_vp, _vp],
    "sphhip_use_default_stream": [_vp],
    "sphhip_get_stream": [_vp, ctypes.POINTER(_vp)],
    "sphhip_synchronize": [_vp],
    "sphhip_set_tuning": [_vp, _ci, _ci],
    "sphhip_upload": [_vp, _vp, _ci, _ci],
    "sphhip_download": [_vp, _vp, _ci],
    "sphhip_download_ids": [_vp, _vp, _vp, _ci],
    "sphhip_count": [_vp, _pi],
    "sphhip_build_grid": [_vp],
    "sphhip_density": [_vp],
    "sphhip_forces": [_vp, _cf, _cf, _cf],
    "sphhip_integrate": [_vp],
    "sphhip_step": [_vp, _cf, _cf, _cf, _ci],
    "sphhip_step_static": [ctypes.POINTER(SphParams), _ci, _vp, _vp, _ci, _cf, _cf, _cf, _ci, _vp],
    "sphhip_step_begin": [_vp, _cf, _cf, _cf, _ci, _ci],
    "sphhip_step_finish": [_vp],
    "sphhip_step_reference_order": [_vp, _cf, _cf, _cf, _ci],
    "sphhip_get_cells": [_vp, _vp, _vp, _vp, _vp],
    "sphhip_get_neighbor_counts": [_vp, _vp, _ci],
    "sphhip_set_owned_columns": [_vp, _ci, _ci],
    "sphhip_pack_halo": [_vp, _ci, _ci, _vp, _ci, _pi, _vp, _ci, _pi],
    "sphhip_pack_halo_async": [_vp, _ci, _ci, _vp, _ci, _vp, _ci],
    "sphhip_append_message": [_vp, _vp, _ci],
    "sphhip_append_messages": [_vp, _vp, _ci, _vp, _ci],
    "sphhip_sync_counts": [_vp, _pi],
    "sphhip_keep_columns": [_vp, _ci, _ci],
    "sphhip_drop_ghosts": [_vp],
    "sphhip_append": [_vp, _vp, _ci],
    "sphhip_column_histogram": [_vp, _vp],
    "sphhip_selftest_division": [_ci, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)],
    "sphhip_set_profiling": [_vp, _ci],
    "sphhip_set_profiling_mask": [_vp, ctypes.c_uint],
    "sphhip_set_profiling_period": [_vp, _ci],
    "sphhip_get_phase_times": [_vp, _vp, _vp],
    "sphhip_reset_phase_times": [_vp],
    "sphhip_get_pair_count": [_vp, ctypes.POINTER(ctypes.c_longlong)],
    "sphhip_get_tile_stats": [_vp, _pi],
}


class SphHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads libsphhip.so (built in-tree by `make -C sph_amd/csrc`); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SphHipError(
            "sph_amd/libsphhip.so is missing: build it with `make -C sph_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback."
        )
    # One HIP runtime per process: torch wheels bundle their own libamdhip64 / libhsa-runtime64.  If libsphhip.so is
    # loaded first it brings in /opt/rocm's copies, torch later adds its own, and whichever runtime initialises second
    # finds no device (hipErrorNoDevice).  Loaded after torch, libsphhip.so binds to the runtime torch already mapped.
    # So when torch is installed it is imported first (SPHHIP_NO_TORCH_PRELOAD=1 skips this: processes that never
    # touch torch).
    if "torch" not in sys.modules and os.environ.get("SPHHIP_NO_TORCH_PRELOAD") != "1":
        import importlib.util

        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.sphhip_last_error.restype = ctypes.c_char_p
    lib.sphhip_last_error.argtypes = []
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().sphhip_last_error()
        raise SphHipError("sphhip error %d: %s" % (rc, msg.decode() if msg else "?"))


def derive(params):
    d = SphDerived()
    check(load().sphhip_derive(ctypes.byref(params), ctypes.byref(d)))
    return d


def default_params():
    p = SphParams()
    check(load().sphhip_default_params(ctypes.byref(p)))
    return p
