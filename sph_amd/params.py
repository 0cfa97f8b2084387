"""ctypes mirrors of the POD structs in include/sphhip.h and the reference's particle record.

Particle layout follows the reference's AoS `struct Particle` (SPH/sph.h:27-33): 28 bytes,
pos @0, vel @8, density @16, pressure @20, valid(bool) @24 + 3 pad bytes.
"""
import ctypes

import numpy as np

PARTICLE_DTYPE = np.dtype(
    {
        "names": ["pos", "vel", "density", "pressure", "valid"],
        "formats": [("<f4", (2,)), ("<f4", (2,)), "<f4", "<f4", "u1"],
        "offsets": [0, 8, 16, 20, 24],
        "itemsize": 28,
    }
)

# 20-byte halo / migration record {x, y, vx, vy, idv}; idv = id | (invalid << 31)
HALO_DTYPE = np.dtype(
    {
        "names": ["pos", "vel", "idv"],
        "formats": [("<f4", (2,)), ("<f4", (2,)), "<u4"],
        "offsets": [0, 8, 16],
        "itemsize": 20,
    }
)


class SphParams(ctypes.Structure):
    """include/sphhip.h `SphParams`; defaults = SPH/sph.h:11-20 + literals of SPH/sph.cpp."""

    _fields_ = [
        ("h", ctypes.c_float),
        ("dt", ctypes.c_float),
        ("mass", ctypes.c_float),
        ("viscosity", ctypes.c_float),
        ("stiffness", ctypes.c_float),
        ("rest_density", ctypes.c_float),
        ("damping", ctypes.c_float),
        ("max_vel", ctypes.c_float),
        ("sigma", ctypes.c_float),
        ("epsilon", ctypes.c_float),
        ("gravity", ctypes.c_float),
        ("force_cap", ctypes.c_float),
        ("mouse_radius", ctypes.c_float),
        ("xmin", ctypes.c_float),
        ("xmax", ctypes.c_float),
        ("ymin", ctypes.c_float),
        ("ymax", ctypes.c_float),
        ("cell_size", ctypes.c_float),
    ]

    @classmethod
    def reference(cls):
        """The reference's compile-time constants (SPH/sph.h:11-20; sph.cpp:77,81,91,113-128)."""
        f = np.float32
        return cls(
            h=f(0.15), dt=f(0.0005), mass=f(0.05), viscosity=f(0.3), stiffness=f(1000.0),
            rest_density=f(1000.0), damping=f(0.9), max_vel=f(5.0), sigma=f(0.05), epsilon=f(1.0),
            gravity=f(10.0), force_cap=f(1000.0), mouse_radius=f(0.25),
            xmin=f(0.05), xmax=f(1.95), ymin=f(0.05), ymax=f(1.95), cell_size=f(0.0),
        )

    def copy(self, **kw):
        out = SphParams()
        ctypes.memmove(ctypes.byref(out), ctypes.byref(self), ctypes.sizeof(self))
        for k, v in kw.items():
            setattr(out, k, np.float32(v))
        return out


class SphDerived(ctypes.Structure):
    """include/sphhip.h `SphDerived`."""

    _fields_ = [
        ("h2", ctypes.c_float),
        ("poly6", ctypes.c_float),
        ("spiky", ctypes.c_float),
        ("visc_lap", ctypes.c_float),
        ("visc_mass", ctypes.c_float),
        ("lj_cut", ctypes.c_float),
        ("lj_24eps", ctypes.c_float),
        ("neg_mass", ctypes.c_float),
        ("mouse_r2", ctypes.c_float),
        ("cell", ctypes.c_float),
        ("inv_cell", ctypes.c_float),
        ("nx", ctypes.c_int),
        ("ny", ctypes.c_int),
    ]
