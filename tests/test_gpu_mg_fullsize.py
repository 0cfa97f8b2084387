"""The slab split of BASELINE configs[3] / configs[4] at FULL size through the C++ driver (libsph_mg.so), every slab
on the one GPU of the test box (device-to-device copies as the transport): the slab boundaries, the per-boundary
message sizes, the hand-over after a rebalance and the particle capacities are those of the 8-GPU / 2-GPU run; only
the wires differ.  Replaces the index blocks of SPH/MPI_sph.cpp:250-257,404-411.  Results must equal the single-GPU
run bit for bit (and the oracle's cell-list step, itself pinned to the compiled reference, for one step)."""
import numpy as np
import pytest

import orc
from sph_amd import lib as _lib
from sph_amd import scenes
from sph_amd.slab import HALO_COLS, grid_column

pytestmark = pytest.mark.gpu


def _single(params, a, nsteps):
    from sph_amd.solver import Solver

    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=nsteps)
        return s.download()


def _halo_fill(params, a, bounds):
    """Records the fullest halo message of the state `a` holds under `bounds` (the HALO_COLS columns on either side of
    every inner boundary)."""
    drv = _lib.derive(params)
    hist = np.bincount(grid_column(a["pos"][:, 0], params, drv), minlength=drv.nx)
    fill = 0
    for b in bounds[1:-1]:
        fill = max(fill, int(hist[b - HALO_COLS:b].sum()), int(hist[b:b + HALO_COLS].sum()))
    return fill


@pytest.mark.parametrize("cols,world,label", [(4096, 8, "configs[3]: 16M dam break over 8 slabs"),
                                              (2896, 2, "configs[4]: 8M per GPU ... the per-GPU scene over 2 slabs")])
def test_full_size_slab_split_bit_exact(cols, world, label):
    from sph_amd.mg import MultiGpuSolver

    params, a = scenes.dam_break(cols)
    n = len(a)
    with MultiGpuSolver(params, a, devices=[0] * world) as m:
        assert m.transport == "copy"
        b0, c0 = m.bounds(), m.counts()
        assert sum(c0) == n and max(c0) - min(c0) <= 2 * 4 * cols  # balanced to the granularity of a grid column
        m.step(1)
        m.synchronize()
        one = m.download()
        m.step(2, rebalance_every=1)  # a rebalance between the two steps
        m.synchronize()  # (raises if a halo message or a particle capacity overflowed)
        got = m.download()
        hs = m.halo_stats()
        assert m.rebalances == 1 and sum(m.counts()) == n
        b1 = m.bounds()
    fill = _halo_fill(params, got, b1)
    print("%s: bounds %s -> %s, fullest halo message %d records of %d capacity (%.0f %%), %.0f bytes per slab and exchange"
          % (label, b0, b1, fill, hs["max_message_records"], 100.0 * fill / hs["max_message_records"], hs["bytes_per_exchange_per_slab"]))
    assert fill <= hs["max_message_records"]
    assert orc.same_bits(one, orc.Oracle().step_cells(params, a.copy(), 1))
    assert orc.same_bits(got, _single(params, a, 3))


def test_soak_four_slabs_of_the_1m_dam_break():
    """4 slabs x 1M-particle dam break pushed to the right at 2 m/s, 240 steps, a rebalance every 60: the boundaries
    follow the fluid, every rebalance hands columns over, the messages keep their per-boundary size in between."""
    from sph_amd.mg import MultiGpuSolver
    from sph_amd.solver import Solver

    params, a = scenes.dam_break(1024)
    a["vel"][:, 0] = 2.0
    with Solver(params, n_max=len(a)) as s, MultiGpuSolver(params, a, devices=[0] * 4) as m:
        s.upload(a)
        b0 = m.bounds()
        for _ in range(2):
            s.stepSimulation(nsteps=120)
            m.step(120, rebalance_every=60)
            m.synchronize()
            assert orc.same_bits_nan(m.download(), s.download())
        assert m.rebalances >= 2 and sum(m.counts()) == len(a)
        print("bounds %s -> %s after %d rebalances; %s" % (b0, m.bounds(), m.rebalances, m.halo_stats()))
