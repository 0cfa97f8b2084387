"""The documented tuning switches of libsphhip.so change speed only: every one of
them is run here against the oracle, bit for bit, on a lattice scene (sparse cells), the dense block of golden
F5 (pressure > 0, long stencil ranges) and a dam-break miniature -- through the fused step and through the
stand-alone phases -- and on a 2-rank slab run (the switches that touch the halo pack / ghost drop).

  SPHHIP_GRID=radix     LDS-local LSD radix sort + suffix-min cell starts instead of the counting sort
  SPHHIP_USE_MASKS=0    k_force scans the candidates itself instead of using k_density's in-range masks
  SPHHIP_FAST_DIV=0     hipcc's generic `/` and sqrtf instead of exact_div.h
  SPHHIP_SMALL_GRID=0   never the one-workgroup grid build
  SPHHIP_LAZY_PACK=0    the halo pack always compacts the state
  SPHHIP_TILE_LIST_FUSED=0  the pair kernels' work list from a launch of its own instead of k_cell_scatter's head workgroups
  SPHHIP_ZERO_P=0       k_force keeps the pressure term (and the spiky gradient) on tiles whose pressures are all +0
"""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

import orc
from sph_amd import scenes
from sph_amd.solver import Solver

pytestmark = pytest.mark.gpu

O = orc.Oracle()
P = O.default_params()

SWITCHES = [("SPHHIP_GRID", "radix"), ("SPHHIP_USE_MASKS", "0"), ("SPHHIP_FAST_DIV", "0"), ("SPHHIP_SMALL_GRID", "0"),
            ("SPHHIP_LAZY_PACK", "0"), ("SPHHIP_TILE_LIST_FUSED", "0"), ("SPHHIP_ZERO_P", "0")]


def _scenes():
    g5 = np.load(os.path.join(orc.GOLDEN, "f5_dense_block.npz"), allow_pickle=False)["state"]
    dam_p, dam = scenes.dam_break(48)
    # lattice6000: about 35 particles per cell of the reference box, stencil ranges of 257..448 candidates ("long" tiles)
    return [("lattice4096", P, scenes.lattice(4096)), ("lattice6000", P, scenes.lattice(6000)), ("f5_dense", P, g5.copy()),
            ("dam48", dam_p, dam)]


@pytest.mark.parametrize("var,val", SWITCHES)
def test_switch_keeps_every_bit(monkeypatch, var, val):
    monkeypatch.setenv(var, val)
    for name, params, a in _scenes():
        with Solver(params, n_max=len(a)) as s:
            # fused step
            s.upload(a)
            s.stepSimulation(nsteps=3)
            got = s.download()
            assert orc.same_bits(got, O.step_cells(params, a.copy(), 3)), (var, name, "step")
            # cell list + neighbour counts
            s.upload(a)
            s.buildNeighbors()
            keys, order, cs, ce = s.cells()
            _, oorder, oskeys, ocs = O.build_cells(params, a)
            assert np.array_equal(keys, oskeys) and np.array_equal(order, oorder), (var, name, "cells")
            assert np.array_equal(cs, ocs[:-1]) and np.array_equal(ce, ocs[1:]), (var, name, "cell table")
            s.computeDensityPressure()
            d, cnt = O.density_cells(params, a.copy())
            assert np.array_equal(s.neighbor_counts(), cnt), (var, name, "counts")
            assert orc.same_bits(s.download(), d), (var, name, "density")
            # force pass on its own (no masks from a density pass of the same grid build: the scan path)
            s.upload(d)
            s.computeForces((1.0, 1.0), 50000.0)
            assert orc.same_bits(s.download(), O.forces_cells(params, d.copy(), (1.0, 1.0), 50000.0)), (var, name, "forces")


def _mixed_pressure_scene():
    """A dam-break miniature (zero pressure everywhere: densities of ~100) with a block at a tenth of the spacing on top
    of it (densities above REST_DENSITY, P > 0): tiles of both kinds in one launch, and tiles that hold both."""
    params, a = scenes.dam_break(64)
    blk = scenes._block(30, 30, np.float32(params.h / 40.0), np.float32(params.xmin + 1.0), np.float32(params.ymin + 2.45))
    b = np.zeros(len(a) + len(blk), a.dtype)
    b[:len(a)] = a
    b[len(a):] = blk
    b["vel"][::7, 0] = 0.25
    b["vel"][::5, 1] = -0.5
    return params, b


@pytest.mark.parametrize("zero_p", ["1", "0"])
def test_zero_pressure_shortcut_on_a_mixed_scene(monkeypatch, zero_p):
    """k_force leaves the pressure term (and the spiky gradient it alone uses) out on tiles staged with every pressure
    +0: bit for bit the same sum (pairs.hip: force_terms<ZEROP>).  Checked where both kinds of tile meet."""
    monkeypatch.setenv("SPHHIP_ZERO_P", zero_p)
    params, a = _mixed_pressure_scene()
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.computeDensityPressure()
        d = s.download()
        n_p = int((d["pressure"] > 0).sum())
        assert 100 < n_p < len(a) // 4, n_p  # some pressure, mostly none
        assert orc.same_bits(d, O.density_cells(params, a.copy())[0])
        # bench.py's report of the share of tiles on each path (host recount over the cell table)
        import bench

        paths = bench.pressure_paths(s, params)
        assert paths["particles_with_nonzero_pressure"] == n_p and 0.0 < paths["zero_pressure_tile_frac"] < 1.0, paths
        for nsteps in (1, 6):
            s.upload(a)
            s.stepSimulation(nsteps=nsteps)
            assert orc.same_bits_nan(s.download(), O.step_cells(params, a.copy(), nsteps)), nsteps


@pytest.mark.parametrize("var,val", [("SPHHIP_LAZY_PACK", "0"), ("SPHHIP_GRID", "radix"), ("SPHHIP_USE_MASKS", "0"),
                                     ("SPH_SLAB_OVERLAP", "0"), ("SPHHIP_SPLIT_CONCURRENT", "0")])
def test_switch_on_two_slabs(monkeypatch, tmp_path, var, val):
    """Two ranks (sharing the box's one GPU, messages staged through gloo), 30 steps with a rebalance, several
    steps per call so that the overlapped step_begin / pack / step_finish path runs: same bits as one GPU."""
    import test_gpu_slab as T

    monkeypatch.setenv(var, val)
    out = str(tmp_path / "got.npy")
    mp.spawn(T._worker, args=(2, T._free_port(), 30, out), nprocs=2, join=True)
    got = np.load(out)
    params, a = T._scene()
    assert orc.same_bits(got, O.step_cells(params, a.copy(), 30)), var


def test_pack_between_the_halves_of_a_step(monkeypatch):
    """sphhip_pack_halo_async between sphhip_step_begin and sphhip_step_finish: allowed only when it can take its
    boundary-only path on columns the first half finished; anything else is SPHHIP_ESTATE, not a silently
    corrupted state."""
    import torch

    from sph_amd.lib import SphHipError

    params, a = scenes.dam_break(96, 64)
    cap = 8192
    dev = torch.device("cuda", 0)
    msg = torch.zeros((cap + 1, 5), dtype=torch.int32, device=dev)
    with Solver(params, n_max=2 * len(a)) as s:
        lo, hi = 10, 40
        s.set_owned_columns(lo, hi)
        s.upload(a)
        s.keep_columns(lo - 2, hi + 2)
        # edge widths below SPHHIP_PACK_REACH on a side with a message
        s.stepBegin(edge_left=1, edge_right=3)
        with pytest.raises(SphHipError, match="-3"):
            s.pack_halo_async(lo + 2, hi - 2, msg.data_ptr(), cap, msg.data_ptr(), cap)
        s.stepFinish()
        # send range wider than the halo: the compacting split may not run mid-step
        s.stepBegin(edge_left=3, edge_right=3)
        with pytest.raises(SphHipError, match="-3"):
            s.pack_halo_async(lo + 5, hi - 2, msg.data_ptr(), cap, None, 0)
        s.stepFinish()
        s.synchronize()
    monkeypatch.setenv("SPHHIP_LAZY_PACK", "0")
    with Solver(params, n_max=2 * len(a)) as s:  # the whole pass runs in step_begin: any pack is fine afterwards
        s.set_owned_columns(lo, hi)
        s.upload(a)
        s.keep_columns(lo - 2, hi + 2)
        s.stepBegin(edge_left=3, edge_right=3)
        s.pack_halo_async(lo + 2, hi - 2, msg.data_ptr(), cap, None, 0)
        s.stepFinish()
        s.sync_counts()


def test_dense_list_overflow_is_reported(monkeypatch):
    """An overflow of the dense sub-tile list (its particles get no update) must surface at the next
    synchronising call on a single domain too (the sticky error word was only read on slabs)."""
    from sph_amd.lib import SphHipError

    monkeypatch.setenv("SPHHIP_DENSE_CAP", "1")
    a = scenes.lattice(16384)  # the reference box at ~100 particles per cell: dense sub-tiles everywhere
    with Solver(P, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=1)
        with pytest.raises(SphHipError, match="-4"):
            s.sync_counts()
        s.upload(a)  # the flag was cleared: the context stays usable
        s.synchronize()
