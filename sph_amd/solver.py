"""Host-side mirror of the reference's step interface (SPH/sph.h:35-39) over libsphhip.so.

`Solver` keeps the particle state resident on one MI355X; the method names follow the reference's
free functions (initSimulation / computeDensityPressure / computeForces / integrate /
stepSimulation) plus the neighbour-search phase the reference only hints at (SPH/sph.cuh:33-34).
Host arrays are numpy structured arrays with the reference's 28-byte AoS `Particle` layout.
"""
import ctypes

import numpy as np

from . import lib as _lib
from .params import PARTICLE_DTYPE, SphParams


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Solver:
    def __init__(self, params=None, n_max=1 << 20, device=0):
        self.L = _lib.load()
        self.params = params if params is not None else SphParams.reference()
        self.drv = _lib.derive(self.params)
        self.n_max = int(n_max)
        h = ctypes.c_void_p()
        _lib.check(self.L.sphhip_create(ctypes.byref(self.params), self.n_max, int(device), ctypes.byref(h)))
        self.h = h

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.sphhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state ------------------------------------------------------------------------------
    @property
    def n(self):
        v = ctypes.c_int()
        _lib.check(self.L.sphhip_count(self.h, ctypes.byref(v)))
        return v.value

    def upload(self, particles, id_base=0):
        a = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        _lib.check(self.L.sphhip_upload(self.h, _ptr(a), len(a), int(id_base)))

    def download(self, with_ids=False):
        n = self.n
        out = np.zeros(n, PARTICLE_DTYPE)
        if with_ids:
            ids = np.zeros(n, np.int32)
            _lib.check(self.L.sphhip_download_ids(self.h, _ptr(out), _ptr(ids), n))
            return out, ids
        _lib.check(self.L.sphhip_download(self.h, _ptr(out), n))
        return out

    # -- phases (names of SPH/sph.h:35-39) -----------------------------------------------------
    def buildNeighbors(self):
        _lib.check(self.L.sphhip_build_grid(self.h))

    def computeDensityPressure(self):
        _lib.check(self.L.sphhip_density(self.h))

    def computeForces(self, mousePos=(0.0, 0.0), interactionStrength=0.0):
        _lib.check(self.L.sphhip_forces(self.h, mousePos[0], mousePos[1], interactionStrength))

    def integrate(self):
        _lib.check(self.L.sphhip_integrate(self.h))

    def stepSimulation(self, mousePos=(0.0, 0.0), interactionStrength=0.0, nsteps=1):
        _lib.check(self.L.sphhip_step(self.h, mousePos[0], mousePos[1], interactionStrength, int(nsteps)))

    def stepBegin(self, mousePos=(0.0, 0.0), interactionStrength=0.0, edge_left=0, edge_right=0):
        """First half of a step (sphhip_step_begin): everything but the force pass of the interior columns."""
        _lib.check(self.L.sphhip_step_begin(self.h, mousePos[0], mousePos[1], interactionStrength, edge_left, edge_right))

    def stepFinish(self):
        _lib.check(self.L.sphhip_step_finish(self.h))

    def stepSimulationReferenceOrder(self, mousePos=(0.0, 0.0), interactionStrength=0.0, nsteps=1):
        """serial_stepSimulation (SPH/sph.cpp:166-171) with the reference's own all-pairs order and in-place
        velocity update: sequential parity tool for n <= 4096, bit-exact against the reference's serial path."""
        _lib.check(self.L.sphhip_step_reference_order(self.h, mousePos[0], mousePos[1], interactionStrength, int(nsteps)))

    def synchronize(self):
        _lib.check(self.L.sphhip_synchronize(self.h))

    # -- diagnostics ------------------------------------------------------------------------
    def cells(self):
        """-> keys[n], order[n] (particle id per sorted slot), cell_start[ncell], cell_end[ncell]"""
        n, nc = self.n, self.drv.nx * self.drv.ny
        keys = np.zeros(n, np.int32)
        order = np.zeros(n, np.int32)
        cs = np.zeros(nc, np.int32)
        ce = np.zeros(nc, np.int32)
        _lib.check(self.L.sphhip_get_cells(self.h, _ptr(keys), _ptr(order), _ptr(cs), _ptr(ce)))
        return keys, order, cs, ce

    def neighbor_counts(self):
        n = self.n
        c = np.zeros(n, np.int32)
        _lib.check(self.L.sphhip_get_neighbor_counts(self.h, _ptr(c), n))
        return c

    def tile_stats(self):
        """-> dict: work-list entries, rows per base tile, long entries, dense entries (density, force) of the last launches."""
        st = (ctypes.c_int * 5)()
        _lib.check(self.L.sphhip_get_tile_stats(self.h, st))
        return dict(zip(("tiles", "tile_rows", "long", "dense_density", "dense_force"), [int(x) for x in st]))

    def pair_count(self):
        v = ctypes.c_longlong()
        _lib.check(self.L.sphhip_get_pair_count(self.h, ctypes.byref(v)))
        return v.value

    def set_tuning(self, tile_cells=0, pair_cap=0):
        _lib.check(self.L.sphhip_set_tuning(self.h, int(tile_cells), int(pair_cap)))

    def set_stream(self, stream_ptr):
        """Run on a caller-owned HIP stream (its handle as an int).  None selects the context's own stream, exactly
        as sphhip_set_stream(ctx, NULL) does.  The handle 0 is HIP's null stream, which the C call cannot express
        through its argument: it is refused here, use use_default_stream()."""
        if stream_ptr is None:
            _lib.check(self.L.sphhip_set_stream(self.h, ctypes.c_void_p(None)))
        elif int(stream_ptr) == 0:
            raise ValueError("stream handle 0 is the null stream: call use_default_stream() (None = the context's own stream)")
        else:
            _lib.check(self.L.sphhip_set_stream(self.h, ctypes.c_void_p(int(stream_ptr))))

    def use_default_stream(self):
        """Run on the device's default (null) stream (sphhip_use_default_stream)."""
        _lib.check(self.L.sphhip_use_default_stream(self.h))

    def follow_torch_stream(self, torch_stream):
        """Run on a torch.cuda stream (torch's default stream is HIP's null stream)."""
        h = int(torch_stream.cuda_stream)
        if h == 0:
            self.use_default_stream()
        else:
            self.set_stream(h)

    def set_profiling(self, on):
        _lib.check(self.L.sphhip_set_profiling(self.h, 1 if on else 0))

    def set_profiling_phases(self, names):
        """Bracket only the named phases (sph_amd.lib.PH_NAMES) with HIP events."""
        mask = 0
        for nm in names:
            mask |= 1 << _lib.PH_NAMES.index(nm)
        _lib.check(self.L.sphhip_set_profiling_mask(self.h, mask))

    def set_profiling_period(self, period):
        """Bracket the selected phases in every `period`-th step only (an event pair costs the stream ~11 us)."""
        _lib.check(self.L.sphhip_set_profiling_period(self.h, int(period)))

    def reset_phase_times(self):
        _lib.check(self.L.sphhip_reset_phase_times(self.h))

    def phase_times(self):
        ms = (ctypes.c_float * len(_lib.PH_NAMES))()
        k = (ctypes.c_int * len(_lib.PH_NAMES))()
        _lib.check(self.L.sphhip_get_phase_times(self.h, ms, k))
        return {name: (float(ms[i]), int(k[i])) for i, name in enumerate(_lib.PH_NAMES)}

    # -- slab hooks -------------------------------------------------------------------------
    def set_owned_columns(self, lo, hi):
        _lib.check(self.L.sphhip_set_owned_columns(self.h, int(lo), int(hi)))

    def pack_halo(self, send_hi_left, send_lo_right, d_left, cap_left, d_right, cap_right):
        nl, nr = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.L.sphhip_pack_halo(self.h, int(send_hi_left), int(send_lo_right), ctypes.c_void_p(d_left),
                                           int(cap_left), ctypes.byref(nl), ctypes.c_void_p(d_right), int(cap_right),
                                           ctypes.byref(nr)))
        return nl.value, nr.value

    def pack_halo_async(self, send_hi_left, send_lo_right, d_msg_left, cap_left, d_msg_right, cap_right):
        _lib.check(self.L.sphhip_pack_halo_async(self.h, int(send_hi_left), int(send_lo_right), ctypes.c_void_p(d_msg_left),
                                                 int(cap_left), ctypes.c_void_p(d_msg_right), int(cap_right)))

    def append_message(self, d_msg, cap):
        _lib.check(self.L.sphhip_append_message(self.h, ctypes.c_void_p(d_msg), int(cap)))

    def append_messages(self, d_msg_a, cap_a, d_msg_b, cap_b):
        """Both neighbours' messages in one launch (a first, b behind it); either may be None."""
        _lib.check(self.L.sphhip_append_messages(self.h, ctypes.c_void_p(d_msg_a), int(cap_a), ctypes.c_void_p(d_msg_b), int(cap_b)))

    def sync_counts(self):
        v = ctypes.c_int()
        _lib.check(self.L.sphhip_sync_counts(self.h, ctypes.byref(v)))
        return v.value

    def keep_columns(self, lo, hi):
        _lib.check(self.L.sphhip_keep_columns(self.h, int(lo), int(hi)))

    def drop_ghosts(self):
        _lib.check(self.L.sphhip_drop_ghosts(self.h))

    def append(self, d_records, n):
        _lib.check(self.L.sphhip_append(self.h, ctypes.c_void_p(d_records), int(n)))

    def column_histogram(self):
        h = np.zeros(self.drv.nx, np.int32)
        _lib.check(self.L.sphhip_column_histogram(self.h, _ptr(h)))
        return h
