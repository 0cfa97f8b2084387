"""The multi-rank slab path (sph_amd/slab.py) on CPU: world_size 2 and 3 under gloo, physics by the
oracle-backed fake backend.  Checks that the slab-decomposed run equals the single-domain run bit
for bit (same canonical summation order), that no particle is lost or duplicated while particles
migrate across slab boundaries, and that rebalancing keeps ownership consistent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from sph_amd import lib as _lib
from sph_amd import scenes
from sph_amd.slab import SlabSolver, balanced_bounds, distribute, grid_column, handover_caps, steady_caps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scene, nsteps, rebalance_every, out_path, overlap="1", skew=False):
    os.environ["SPH_SLAB_OVERLAP"] = overlap
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_backend import FakeSlabBackend

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params, a = _scene(scene)
        backend = FakeSlabBackend(params)
        hist = np.bincount(grid_column(a["pos"][:, 0], params, backend.drv), minlength=backend.drv.nx)
        bounds = balanced_bounds(hist, world)
        if skew:  # start far from balance: a rebalance moves a boundary by many columns at once
            import sph_amd.slab as slab_mod

            slab_mod.HALO_FLOOR = 16
            bounds = [0] + [3 * g for g in range(1, world)] + [backend.drv.nx]
        slab = SlabSolver(backend, params, bounds, hist=hist)
        distribute(slab, a)
        # several steps per call: between the steps of one call the exchange is overlapped with the step
        # (step_begin / pack / step_finish), the first exchange of a call is not
        done = 0
        while done < nsteps:
            if rebalance_every and done:
                slab.rebalance()
            k = min(rebalance_every or nsteps, nsteps - done)
            slab.step(k)
            done += k
        got = slab.gather(len(a))
        if rank == 0:
            np.save(out_path, got)
            np.save(out_path + ".bounds.npy", np.array(slab.bounds))
    finally:
        dist.destroy_process_group()


def _scene(name):
    if name == "dam":
        return scenes.dam_break(40, 24)
    if name == "sloshing":  # fluid with a strong x-velocity: many particles cross slab boundaries
        p, a = scenes.dam_break(36, 20)
        a["vel"][:, 0] = 4.0
        a["pos"][:, 0] += np.float32(2.0)
        return p, a
    if name == "wide":  # 96 x 24: 24 grid columns of ~96 particles
        p, a = scenes.dam_break(96, 24)
        a["vel"][:, 0] = 2.0
        return p, a
    raise KeyError(name)


@pytest.mark.parametrize("world,scene,nsteps,rebalance_every,overlap", [(2, "dam", 6, 0, "1"), (3, "sloshing", 40, 0, "1"),
                                                                         (2, "sloshing", 30, 7, "1"),
                                                                         (2, "sloshing", 30, 7, "0"), (2, "dam", 5, 1, "1")])
def test_slab_equals_single_domain(tmp_path, world, scene, nsteps, rebalance_every, overlap):
    out = str(tmp_path / "got.npy")
    mp.spawn(_worker, args=(world, _free_port(), scene, nsteps, rebalance_every, out, overlap), nprocs=world, join=True)
    got = np.load(out)
    params, a = _scene(scene)
    want = orc.Oracle().step_cells(params, a.copy(), nsteps)
    assert orc.same_bits(got, want)


def test_large_boundary_shift_fits_the_handover_message(tmp_path):
    """A rebalance may move a boundary by almost a whole neighbour slab; the first exchange afterwards hands all those
    columns over in one message, which is sized for it from the histogram (handover_caps).  4 slabs that start 3
    columns wide each (all the rest with the last rank), small column populations (24 rows: no floor hides anything --
    the fake backend fails on the first record beyond a message's capacity)."""
    out = str(tmp_path / "got.npy")
    world, nsteps = 4, 12
    mp.spawn(_worker, args=(world, _free_port(), "wide", nsteps, 3, out, "1", True), nprocs=world, join=True)
    got = np.load(out)
    params, a = _scene("wide")
    want = orc.Oracle().step_cells(params, a.copy(), nsteps)
    assert orc.same_bits(got, want)
    b = np.load(out + ".bounds.npy")
    assert b[3] > 15  # (from 9: boundaries move one neighbour's width at a time, the last one first)


def _failing_worker(rank, world, port, out_dir):
    """Rank 1's backend reports an overflow at its next synchronising call (as the library does: sticky device flag,
    raised by the histogram download / sync_counts); rank 0 is healthy."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import datetime

    from fake_backend import FakeSlabBackend

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        params, a = _scene("dam")
        backend = FakeSlabBackend(params)
        hist = np.bincount(grid_column(a["pos"][:, 0], params, backend.drv), minlength=backend.drv.nx)
        slab = SlabSolver(backend, params, balanced_bounds(hist, world), hist=hist)
        distribute(slab, a)
        slab.step(2)
        if rank == 1:
            def boom():
                raise RuntimeError("halo message overflow (simulated)")

            backend.column_histogram = boom
            backend.check = boom
        seen = []
        for call in (slab.rebalance, slab.check_all):
            try:
                call()
                seen.append("ok")
            except RuntimeError as e:
                seen.append(str(e))
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
            f.write("\n".join(seen))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_fails_every_rank_instead_of_hanging_them(tmp_path):
    """A rank that finds an overflow at a rebalance (or at check_all) must not leave its neighbours waiting in the
    all-reduce it never joins: the failure travels with the reduced vector and every rank raises."""
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = open(tmp_path / "rank0.txt").read().splitlines()
    r1 = open(tmp_path / "rank1.txt").read().splitlines()
    assert all("other rank(s) reported a failure" in m for m in r0), r0
    assert all("simulated" in m for m in r1), r1


def test_message_sizes_from_the_histogram():
    hist = np.zeros(60, np.int64)
    hist[:40] = 100
    hist[20] = 400  # a full column next to boundary 2
    bounds = [0, 10, 21, 40, 60]
    caps = steady_caps(hist, bounds, margin=1.25, floor=16)
    assert caps[0] == 0 and caps[-1] == 0
    assert caps[1] == int(1.25 * 2 * 100) and caps[2] == int(1.25 * 2 * 400) and caps[3] == int(1.25 * 2 * 100)
    new = [0, 10, 30, 40, 60]  # boundary 2 moves right by 9 columns: columns 19 .. 31 travel
    first = handover_caps(hist, bounds, new, steady_caps(hist, new, margin=1.25, floor=16))
    need = int(hist[19:32].sum())
    assert first[2] >= need and first[1] == steady_caps(hist, new, margin=1.25, floor=16)[1]


def test_balanced_bounds():
    hist = np.zeros(100, np.int64)
    hist[:25] = 40  # dam break: all the fluid in the left quarter
    b = balanced_bounds(hist, 4)
    assert b[0] == 0 and b[-1] == 100 and all(b[i + 1] - b[i] >= 3 for i in range(4))
    loads = [hist[b[i]:b[i + 1]].sum() for i in range(4)]
    assert max(loads) <= 1.3 * hist.sum() / 4
    # clamped moves: every new boundary stays between its old neighbours
    prev = [0, 10, 20, 30, 100]
    nb = balanced_bounds(hist, 4, prev=prev)
    for g in range(1, 4):
        assert prev[g - 1] + 3 <= nb[g] <= prev[g + 1] - 3
    with pytest.raises(ValueError):
        balanced_bounds(np.ones(11), 4)


def test_grid_column_matches_oracle_keys():
    params, a = scenes.dam_break(30, 30)
    O = orc.Oracle()
    keys, _, _, _ = O.build_cells(params, a)
    drv = _lib.derive(params)
    assert np.array_equal(grid_column(a["pos"][:, 0], params, drv), keys // drv.ny)


def test_sharded_scene_equals_global_scene():
    p2, n_total, hist_fn, make_local = scenes.dam_break_sharded(50, 30)
    params, a = scenes.dam_break(50, 30)
    assert n_total == len(a) and p2.xmax == params.xmax and p2.ymax < params.ymax  # box height follows the rows
    params = p2
    ps, _ = scenes.dam_break(40)
    assert bytes(scenes.dam_break_sharded(40, 40)[0]) == bytes(ps)  # square column: the same box as dam_break
    drv = _lib.derive(params)
    hist = hist_fn(drv)
    assert hist.sum() == n_total
    assert np.array_equal(hist, np.bincount(grid_column(a["pos"][:, 0], params, drv), minlength=drv.nx))
    b = balanced_bounds(hist, 3)
    seen = np.zeros(n_total, int)
    for g in range(3):
        la, ids = make_local(b[g], b[g + 1], drv)
        assert orc.same_bits(la, a[ids])
        seen[ids] += 1
    assert np.all(seen == 1)
