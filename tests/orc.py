"""ctypes access to the CPU oracle (oracle/libsph_oracle.so) and, when built, to the reference's
own serial step (oracle/_ref/libsph_ref.so).  Test infrastructure only."""
import ctypes
import os

import numpy as np

from sph_amd.params import PARTICLE_DTYPE, SphParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libsph_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsph_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

GAUSS_SEIDEL = 0
SNAPSHOT = 1
EPS_F32 = 2

_c_f = ctypes.c_float
_c_i = ctypes.c_int
_vp = ctypes.c_void_p
_pp = ctypes.POINTER(SphParams)


def _ptr(a):
    return a.ctypes.data_as(_vp)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/libsph_oracle.so missing: run `make -C oracle`")
        L = ctypes.CDLL(ORACLE_SO)
        self.L = L
        L.orc_default_params.argtypes = [_pp]
        L.orc_poly6.restype = _c_f
        L.orc_poly6.argtypes = [_c_f, _c_f]
        L.orc_spiky_coeff.restype = _c_f
        L.orc_spiky_coeff.argtypes = [_c_f, _c_f, _c_i]
        L.orc_visc_lap.restype = _c_f
        L.orc_visc_lap.argtypes = [_c_f, _c_f]
        L.orc_lj_coeff.restype = _c_f
        L.orc_lj_coeff.argtypes = [_c_f, _c_f, _c_f]
        L.orc_density_allpairs.argtypes = [_pp, _vp, _c_i]
        L.orc_neighbor_counts_allpairs.argtypes = [_pp, _vp, _c_i, _vp]
        L.orc_forces_allpairs.argtypes = [_pp, _vp, _c_i, _c_f, _c_f, _c_f, _c_i]
        L.orc_integrate.argtypes = [_pp, _vp, _c_i]
        L.orc_step_allpairs.argtypes = [_pp, _vp, _c_i, _c_f, _c_f, _c_f, _c_i, _c_i]
        L.orc_init_lattice.argtypes = [_vp, _c_i, _c_f, _c_f, _c_f, _c_i]
        L.orc_grid_dims.argtypes = [_pp, ctypes.POINTER(_c_f), ctypes.POINTER(_c_f), ctypes.POINTER(_c_i),
                                    ctypes.POINTER(_c_i)]
        L.orc_build_cells.argtypes = [_pp, _vp, _c_i, _vp, _vp, _vp, _vp]
        L.orc_density_cells.argtypes = [_pp, _vp, _c_i, _vp]
        L.orc_forces_cells.argtypes = [_pp, _vp, _c_i, _c_f, _c_f, _c_f, _c_i]
        L.orc_step_cells.argtypes = [_pp, _vp, _c_i, _c_f, _c_f, _c_f, _c_i, _c_i]
        L.orc_omp_step_allpairs.argtypes = [_pp, _vp, _c_i, _c_i]
        L.orc_checksum.restype = ctypes.c_ulonglong
        L.orc_checksum.argtypes = [_vp, _c_i]
        L.orc_num_threads.restype = _c_i

    def default_params(self):
        p = SphParams()
        self.L.orc_default_params(ctypes.byref(p))
        return p

    @staticmethod
    def _chk(a):
        assert a.dtype == PARTICLE_DTYPE and a.flags["C_CONTIGUOUS"]
        return a

    def init_lattice(self, n, min_pos=0.05, max_pos=1.95, rest_density=1000.0, sqrt_f32=False):
        a = np.zeros(n, PARTICLE_DTYPE)
        self.L.orc_init_lattice(_ptr(a), n, min_pos, max_pos, rest_density, int(sqrt_f32))
        return a

    def density_allpairs(self, p, a):
        self.L.orc_density_allpairs(ctypes.byref(p), _ptr(self._chk(a)), len(a))
        return a

    def neighbor_counts_allpairs(self, p, a):
        c = np.zeros(len(a), np.int32)
        self.L.orc_neighbor_counts_allpairs(ctypes.byref(p), _ptr(self._chk(a)), len(a), _ptr(c))
        return c

    def forces_allpairs(self, p, a, mouse=(0.0, 0.0), strength=0.0, flags=GAUSS_SEIDEL):
        self.L.orc_forces_allpairs(ctypes.byref(p), _ptr(self._chk(a)), len(a), mouse[0], mouse[1], strength, flags)
        return a

    def integrate(self, p, a):
        self.L.orc_integrate(ctypes.byref(p), _ptr(self._chk(a)), len(a))
        return a

    def step_allpairs(self, p, a, nsteps=1, mouse=(0.0, 0.0), strength=0.0, flags=GAUSS_SEIDEL):
        self.L.orc_step_allpairs(ctypes.byref(p), _ptr(self._chk(a)), len(a), mouse[0], mouse[1], strength, flags,
                                 nsteps)
        return a

    def grid_dims(self, p):
        cell, inv, nx, ny = _c_f(), _c_f(), _c_i(), _c_i()
        self.L.orc_grid_dims(ctypes.byref(p), ctypes.byref(cell), ctypes.byref(inv), ctypes.byref(nx),
                             ctypes.byref(ny))
        return cell.value, inv.value, nx.value, ny.value

    def build_cells(self, p, a):
        """-> keys (orig order), order (slot->index), sorted_keys, cell_start[ncell+1]"""
        _, _, nx, ny = self.grid_dims(p)
        n = len(a)
        keys = np.zeros(n, np.int32)
        order = np.zeros(n, np.int32)
        skeys = np.zeros(n, np.int32)
        cs = np.zeros(nx * ny + 1, np.int32)
        self.L.orc_build_cells(ctypes.byref(p), _ptr(self._chk(a)), n, _ptr(keys), _ptr(order), _ptr(skeys), _ptr(cs))
        return keys, order, skeys, cs

    def density_cells(self, p, a):
        c = np.zeros(len(a), np.int32)
        self.L.orc_density_cells(ctypes.byref(p), _ptr(self._chk(a)), len(a), _ptr(c))
        return a, c

    def forces_cells(self, p, a, mouse=(0.0, 0.0), strength=0.0, flags=SNAPSHOT):
        self.L.orc_forces_cells(ctypes.byref(p), _ptr(self._chk(a)), len(a), mouse[0], mouse[1], strength, flags)
        return a

    def step_cells(self, p, a, nsteps=1, mouse=(0.0, 0.0), strength=0.0, flags=SNAPSHOT):
        self.L.orc_step_cells(ctypes.byref(p), _ptr(self._chk(a)), len(a), mouse[0], mouse[1], strength, flags, nsteps)
        return a

    def omp_step_allpairs(self, p, a, nsteps=1):
        self.L.orc_omp_step_allpairs(ctypes.byref(p), _ptr(self._chk(a)), len(a), nsteps)
        return a

    def checksum(self, a):
        return int(self.L.orc_checksum(_ptr(self._chk(a)), len(a)))

    def num_threads(self):
        return int(self.L.orc_num_threads())


class Reference:
    """The reference's own SPH/sph.cpp, compiled by oracle/Makefile into oracle/_ref."""

    def __init__(self):
        L = ctypes.CDLL(REF_SO)
        self.L = L
        assert L.ref_sizeof_particle() == 28
        L.ref_init.argtypes = [_vp, _c_i]
        L.ref_density.argtypes = [_vp, _c_i]
        L.ref_forces.argtypes = [_vp, _c_i, _c_f, _c_f, _c_f]
        L.ref_integrate.argtypes = [_vp, _c_i]
        L.ref_step.argtypes = [_vp, _c_i, _c_f, _c_f, _c_f, _c_i]
        L.ref_poly6.restype = _c_f
        L.ref_poly6.argtypes = [_c_f, _c_f]
        L.ref_spiky_grad.argtypes = [_c_f, _c_f, _c_f, _c_f, ctypes.POINTER(_c_f), ctypes.POINTER(_c_f)]
        L.ref_visc_lap.restype = _c_f
        L.ref_visc_lap.argtypes = [_c_f, _c_f]
        L.ref_lj.argtypes = [_c_f, _c_f, _c_f, _c_f, _c_f, ctypes.POINTER(_c_f), ctypes.POINTER(_c_f)]
        L.ref_const.restype = _c_f
        L.ref_const.argtypes = [_c_i]

    def init(self, n):
        a = np.zeros(n, PARTICLE_DTYPE)
        self.L.ref_init(_ptr(a), n)
        return a

    def density(self, a):
        self.L.ref_density(_ptr(a), len(a))
        return a

    def forces(self, a, mouse=(0.0, 0.0), strength=0.0):
        self.L.ref_forces(_ptr(a), len(a), mouse[0], mouse[1], strength)
        return a

    def integrate(self, a):
        self.L.ref_integrate(_ptr(a), len(a))
        return a

    def step(self, a, nsteps=1, mouse=(0.0, 0.0), strength=0.0):
        self.L.ref_step(_ptr(a), len(a), mouse[0], mouse[1], strength, nsteps)
        return a

    def spiky_grad(self, rx, ry, r_len, h):
        gx, gy = _c_f(), _c_f()
        self.L.ref_spiky_grad(rx, ry, r_len, h, ctypes.byref(gx), ctypes.byref(gy))
        return gx.value, gy.value

    def lj(self, rx, ry, r_len, sigma, eps):
        gx, gy = _c_f(), _c_f()
        self.L.ref_lj(rx, ry, r_len, sigma, eps, ctypes.byref(gx), ctypes.byref(gy))
        return gx.value, gy.value


def have_ref():
    return os.path.exists(REF_SO)


def bits(a):
    """View the numeric fields of a particle array as raw bytes (24 payload bytes + valid)."""
    a = np.ascontiguousarray(a)
    raw = a.view(np.uint8).reshape(len(a), 28)
    return raw[:, :25]


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def same_bits_nan(a, b):
    """same_bits, except that a NaN matches a NaN whatever its sign and payload (x86 produces the
    negative default NaN, gfx950 the positive one)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if len(a) != len(b) or not np.array_equal(a["valid"], b["valid"]):
        return False
    fa = a.view(np.uint8).reshape(len(a), 28)[:, :24].copy().view(np.float32)
    fb = b.view(np.uint8).reshape(len(b), 28)[:, :24].copy().view(np.float32)
    both_nan = np.isnan(fa) & np.isnan(fb)
    return bool(np.all((fa.view(np.uint32) == fb.view(np.uint32)) | both_nan))
