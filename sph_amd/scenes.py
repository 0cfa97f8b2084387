"""Scene generators (host, numpy, float32 arithmetic in the reference's expression order).

`lattice` is the reference's serial_initSimulation (SPH/sph.cpp:132-164) with the square box as a
parameter; `dam_break` and `uniform_box` are the synthetic scenes of BASELINE.json configs 2-5
(SURVEY.md section 8(d)): all sph.h constants kept, lattice spacing H/4 (about 50 neighbours), walls
scaled with the particle count so that the neighbour count does not grow with N as it does in the
reference's fixed 1.9 x 1.9 box.
"""
import math

import numpy as np

from .params import PARTICLE_DTYPE, SphParams

F = np.float32


def lattice(n, min_pos=0.05, max_pos=1.95, rest_density=1000.0):
    """serial_initSimulation, SPH/sph.cpp:132-164 (bit-exact for the reference box)."""
    a = np.zeros(n, PARTICLE_DTYPE)
    if n == 0:
        return a
    cols = int(math.ceil(math.sqrt(n)))  # (int)ceil(sqrt((double)N)), sph.cpp:136
    rows = (n + cols - 1) // cols
    if rows == 0:
        rows = 1
    mn, mx = F(min_pos), F(max_pos)
    rng = F(mx - mn)
    sx = F(rng / F(cols - 1 if cols > 1 else 1))
    sy = F(rng / F(rows - 1 if rows > 1 else 1))
    sp = sy if sy < sx else sx  # std::min(spacingX, spacingY)
    start_x = F(mn + F(F(rng - F(F(cols - 1) * sp)) / F(2.0)))
    start_y = F(mx - F(F(rng - F(F(rows - 1) * sp)) / F(2.0)))
    i = np.arange(n)
    x = (i % cols).astype(np.float32)
    y = (i // cols).astype(np.float32)
    a["pos"][:, 0] = start_x + x * sp
    a["pos"][:, 1] = start_y - y * sp
    a["density"] = F(rest_density)
    a["valid"] = 1
    return a


def _params_for_box(width, height, base=None):
    p = (base or SphParams.reference()).copy()
    p.xmin = F(0.05)
    p.ymin = F(0.05)
    p.xmax = F(F(0.05) + F(width))
    p.ymax = F(F(0.05) + F(height))
    return p


def _block(cols, rows, spacing, x0, y0, rest_density=1000.0):
    n = cols * rows
    a = np.zeros(n, PARTICLE_DTYPE)
    i = np.arange(n)
    sp = F(spacing)
    a["pos"][:, 0] = F(x0) + (i % cols).astype(np.float32) * sp
    a["pos"][:, 1] = F(y0) + (i // cols).astype(np.float32) * sp
    a["density"] = F(rest_density)
    a["valid"] = 1
    return a


def dam_break(cols, rows=None, spacing=None, base=None):
    """2-D dam break: a cols x rows lattice column of fluid at rest against the left wall and the
    floor of a box 4 W wide and 2 W high (W = cols * spacing).  Deterministic, no RNG.
    Returns (params, particles)."""
    rows = rows or cols
    base = base or SphParams.reference()
    sp = F(spacing) if spacing else F(base.h / F(4.0))
    w = F(F(cols) * sp)
    p = _params_for_box(F(4.0) * w, F(2.0) * w, base)
    a = _block(cols, rows, sp, p.xmin, p.ymin, base.rest_density)
    return p, a


def uniform_box(side, spacing=None, base=None):
    """side x side lattice filling a square box wall to wall, as serial_initSimulation fills the
    reference box (sph.cpp:132-164) but at spacing H/4."""
    base = base or SphParams.reference()
    sp = F(spacing) if spacing else F(base.h / F(4.0))
    width = F(F(side - 1) * sp)
    p = _params_for_box(width, width, base)
    a = _block(side, side, sp, p.xmin, p.ymin, base.rest_density)
    return p, a


def _lattice_sharded(p, cols, rows, sp, base):
    """cols x rows lattice from (xmin, ymin) at spacing sp, generated slab by slab: see dam_break_sharded."""
    xs = (F(p.xmin) + np.arange(cols).astype(np.float32) * sp).astype(np.float32)
    ys = (F(p.ymin) + np.arange(rows).astype(np.float32) * sp).astype(np.float32)

    def lattice_cols(drv):
        f = np.floor((xs - F(p.xmin)) * F(drv.inv_cell))
        return np.clip(f, 0, drv.nx - 1).astype(np.int64)

    def column_hist(drv):
        return np.bincount(lattice_cols(drv), minlength=drv.nx) * rows

    def make_local(c0, c1, drv):
        gc = lattice_cols(drv)
        mine = np.nonzero((gc >= c0) & (gc < c1))[0]
        n = len(mine) * rows
        a = np.zeros(n, PARTICLE_DTYPE)
        cc, rr = np.meshgrid(mine, np.arange(rows))
        a["pos"][:, 0] = xs[cc.ravel()]
        a["pos"][:, 1] = ys[rr.ravel()]
        a["density"] = F(base.rest_density)
        a["valid"] = 1
        ids = rr.ravel().astype(np.int64) * cols + cc.ravel()
        return a, ids

    return p, cols * rows, column_hist, make_local


def dam_break_sharded(cols, rows=None, spacing=None, base=None):
    """The dam-break scene for slab-decomposed runs: returns (params, n_total, column_hist_fn,
    make_local) where make_local(c0, c1, drv) generates only the particles whose grid column is in
    [c0, c1) together with their global ids (id = row * cols + col, as `dam_break` numbers them)."""
    rows = rows or cols
    base = base or SphParams.reference()
    sp = F(spacing) if spacing else F(base.h / F(4.0))
    w = F(F(cols) * sp)
    # box: 4 x the fluid width, 2 x the fluid height (for cols == rows exactly `dam_break`'s box); the
    # height does not grow with the number of slabs, so weak scaling keeps the per-rank grid work fixed
    p = _params_for_box(F(4.0) * w, F(2.0) * F(F(rows) * sp), base)
    return _lattice_sharded(p, cols, rows, sp, base)


def uniform_box_sharded(cols, rows=None, spacing=None, base=None):
    """`uniform_box` for slab-decomposed runs (cols x rows lattice filling its box wall to wall; weak scaling
    widens the box with the rank count): same return value as dam_break_sharded."""
    rows = rows or cols
    base = base or SphParams.reference()
    sp = F(spacing) if spacing else F(base.h / F(4.0))
    p = _params_for_box(F(F(cols - 1) * sp), F(F(rows - 1) * sp), base)
    return _lattice_sharded(p, cols, rows, sp, base)

