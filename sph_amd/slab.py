"""Multi-GPU driver: 1-D spatial slabs along x with ghost-column halo exchange and particle
migration, one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in
the CPU tests).

Replaces the reference's MPI partitioning (SPH/MPI_sph.cpp:135-166,255-257): there every rank
holds ALL particles, owns an index block, and receives 24*N bytes per step through four
MPI_Allgathers.  Here rank g owns the grid columns [bounds[g], bounds[g+1]) -- a contiguous key range
because cell keys are column-major -- and per step exchanges, with its two x-neighbours only:
  * the particles of its two outermost columns on each side (ghosts for the neighbour; two columns
    so that the neighbour can recompute the density of its first ghost column locally, which
    replaces the dens/pres Allgathers of MPI_sph.cpp:163-166), and
  * the particles whose position left the slab (migrants; the neighbour becomes their owner),
in ONE message per direction: 20-byte records {x, y, vx, vy, id|valid}.  No collective is on the
per-step path; rebalancing uses one all-reduce of the per-column histogram (nx ints).

Results are identical to the single-GPU path bit for bit: the sorted order is canonical
(key, id) whatever order particles arrive in, ghosts are sorted into the same cells, and every
owned particle sums the same neighbours in the same order.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib
from .params import HALO_DTYPE
from .solver import Solver

HALO_COLS = 2  # SPHHIP_HALO_COLS
# A slab must be wider than the halo: a particle that enters a neighbour's slab from its far side
# must not land inside my ghost range in the same step (it would reach me one step late).
MIN_SLAB_COLS = HALO_COLS + 1
# The halo messages are packed from the first / last PACK_REACH owned columns (a particle crosses less than
# one column per step: sphhip_pack_halo_async, ctx.hip): what step_begin has to finish before the pack.
PACK_REACH = HALO_COLS + 1
BIG = 0x7FFFFFFF


def grid_column(x, params, drv):
    """Grid column of x-coordinates, same binary32 arithmetic as the device (sph_cell_coord)."""
    x = np.asarray(x, np.float32)
    f = np.floor((x - np.float32(params.xmin)) * np.float32(drv.inv_cell))
    return np.clip(np.nan_to_num(f, nan=0.0), 0, drv.nx - 1).astype(np.int64)


def balanced_bounds(hist, world, min_width=MIN_SLAB_COLS, prev=None):
    """Column boundaries giving every rank about the same particle count.  Every slab is at least
    `min_width` columns wide (a halo never spans more than one neighbour).  With `prev`, boundary g
    stays strictly between prev[g-1] and prev[g+1] so that every particle's new owner is its old
    owner or an adjacent rank (migration is nearest-neighbour only)."""
    hist = np.asarray(hist, np.int64)
    nx = len(hist)
    if nx < world * min_width:
        raise ValueError("grid has %d columns: too few for %d slabs of >= %d columns" % (nx, world, min_width))
    csum = np.concatenate([[0], np.cumsum(hist)])
    total = csum[-1]
    b = [0]
    for g in range(1, world):
        target = total * g / world
        c = int(np.searchsorted(csum, target, side="left"))
        lo = b[-1] + min_width
        hi = nx - (world - g) * min_width
        if prev is not None:
            lo = max(lo, prev[g - 1] + min_width)
            hi = min(hi, prev[g + 1] - min_width)
        b.append(int(min(max(c, lo), hi)))
    b.append(nx)
    return b


HALO_MARGIN = float(os.environ.get("SPH_HALO_MARGIN", "1.25"))  # steady-state message size over the halo's population
HALO_FLOOR = 1024


def steady_caps(hist, bounds, margin=None, floor=None):
    """Records per message across every inner slab boundary between two rebalances (list indexed by boundary g =
    1 .. world-1; entries 0 and world are 0).  A message carries the SPHHIP_HALO_COLS columns next to the boundary
    plus the few migrants of one step; it is sized from the fullest column within reach of the boundary (the halo
    and two columns more on each side: what a front moving at MAX_VEL can bring there in 100 steps), times `margin`.
    Sender and receiver compute the same number from the same global histogram -- that is what lets a message
    travel at a size the host fixes beforehand (no count round trip) while shipping about what is needed, not the
    2.5 x fullest-column-of-the-domain of the first version."""
    margin = HALO_MARGIN if margin is None else margin
    floor = HALO_FLOOR if floor is None else floor
    hist = np.asarray(hist, np.int64)
    caps = [0] * len(bounds)
    for g in range(1, len(bounds) - 1):
        b = bounds[g]
        near = hist[max(b - HALO_COLS - 2, 0):b + HALO_COLS + 2]
        caps[g] = int(max(margin * HALO_COLS * float(near.max() if len(near) else 0), floor))
    return caps


def handover_caps(hist, old, new, steady):
    """Records per message of the FIRST exchange after the boundaries moved from `old` to `new`: that exchange hands
    over every column between the old and the new boundary besides the halo.  The histogram was taken at the
    rebalance itself, so the sum over those columns is what the message will hold (plus 1/16 + 256: migrants that are
    still with their old owner are counted at their new column)."""
    hist = np.asarray(hist, np.int64)
    caps = list(steady)
    for g in range(1, len(new) - 1):
        if old[g] == new[g]:
            continue
        lo, hi = min(old[g], new[g]) - HALO_COLS, max(old[g], new[g]) + HALO_COLS
        need = int(hist[max(lo, 0):hi].sum())
        caps[g] = max(steady[g], need + need // 16 + 256)
    return caps


class HipSlabBackend:
    """The product backend: libsphhip.so on one GPU, exchange messages as torch CUDA tensors.
    A message is an int32 tensor of (cap + 1) x 5 words: row 0 is the header (word 0 = number of
    20-byte records that follow), so it can be sent at its fixed size without a count round trip."""

    def __init__(self, params, n_max, device, halo_cap):
        self.device = torch.device("cuda", device)
        self.solver = Solver(params, n_max=n_max, device=device)
        self.drv = self.solver.drv
        # run the library on torch's current stream so that RCCL ops and kernels are ordered
        self.solver.follow_torch_stream(torch.cuda.current_stream(self.device))
        self.n_max = int(n_max)
        self.cap = 0
        self.reserve(halo_cap)

    def reserve(self, halo_cap):
        """Makes the four message buffers hold at least (halo_cap + 1) 20-byte records each (grow only; called at
        set-up and at a rebalance, never inside a run of steps).  What travels is a prefix of a buffer: the size of
        every exchange is fixed by the SlabSolver per slab boundary (steady_caps / handover_caps)."""
        if int(halo_cap) <= self.cap:
            return
        torch.cuda.synchronize(self.device)
        self.cap = int(halo_cap)
        mk = lambda: torch.zeros((self.cap + 1, 5), dtype=torch.int32, device=self.device)  # noqa: E731
        self.send_l, self.send_r, self.recv_l, self.recv_r = mk(), mk(), mk(), mk()

    def set_owned(self, lo, hi):
        self.solver.set_owned_columns(lo, hi)

    def pack(self, send_hi_left, send_lo_right, cap_left, cap_right):
        """Enqueues the split; returns the two outgoing messages, header + cap records each (None where there is
        no neighbour: cap 0)."""
        assert max(cap_left, cap_right) <= self.cap
        self.solver.pack_halo_async(send_hi_left, send_lo_right, self.send_l.data_ptr() if cap_left else None, cap_left,
                                    self.send_r.data_ptr() if cap_right else None, cap_right)
        return (self.send_l[:cap_left + 1] if cap_left else None), (self.send_r[:cap_right + 1] if cap_right else None)

    def recv_buffers(self, cap_left, cap_right):
        return (self.recv_l[:cap_left + 1] if cap_left else None), (self.recv_r[:cap_right + 1] if cap_right else None)

    def finish(self, cap_left, cap_right):
        """Appends what was received; counts stay on the device (no host synchronisation)."""
        self.solver.append_messages(self.recv_l.data_ptr() if cap_left else None, cap_left,
                                    self.recv_r.data_ptr() if cap_right else None, cap_right)

    def check(self):
        """Synchronises and raises if a halo buffer or the particle capacity overflowed."""
        return self.solver.sync_counts()

    def append_host(self, rec_np):
        t = torch.from_numpy(np.ascontiguousarray(rec_np).view(np.int32).reshape(-1, 5)).to(self.device)
        if t.shape[0]:
            self.solver.append(t.data_ptr(), t.shape[0])
        torch.cuda.synchronize(self.device)

    def step(self, mouse, strength):
        self.solver.stepSimulation(mouse, strength, 1)

    def step_begin(self, mouse, strength, has_left, has_right):
        """Everything of a step that the next halo messages depend on (the columns within PACK_REACH of a
        slab boundary); step_finish() runs the interior."""
        self.solver.stepBegin(mouse, strength, PACK_REACH if has_left else 0, PACK_REACH if has_right else 0)

    def step_finish(self):
        self.solver.stepFinish()

    def drop_ghosts(self):
        self.solver.drop_ghosts()

    def download(self):
        return self.solver.download(with_ids=True)

    def column_histogram(self):
        return self.solver.column_histogram()

    @property
    def n(self):
        return self.solver.n


class SlabSolver:
    """x-slab decomposition of one scene over the ranks of a process group."""

    def __init__(self, backend, params, bounds, group=None, hist=None):
        self.backend = backend
        self.params = params
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.solver = getattr(backend, "solver", None)
        # gloo cannot send CUDA tensors: stage the messages through host memory (tests / debugging);
        # under nccl (RCCL) the device buffers go straight over xGMI
        self.stage_cpu = dist.get_backend(group) == "gloo"
        self._ops_cache = {}
        self.overlap = os.environ.get("SPH_SLAB_OVERLAP", "1") != "0"  # exchange overlapped with the interior columns
        self.last_hist = None
        self.last_spread = None
        self.rebalances = 0
        self.records_sent = 0  # records' worth of bytes this rank has sent (headers included), and in how many exchanges
        self.exchanges = 0
        self.set_bounds(bounds)
        if hist is not None:
            self.last_hist = np.asarray(hist, np.int64)
            self.set_caps(steady_caps(hist, self.bounds))
        else:  # (tests that drive a backend without a histogram: one size for everything)
            self.set_caps([0] + [getattr(backend, "CAP", HALO_FLOOR)] * (self.world - 1) + [0])

    # -- partition ----------------------------------------------------------------------------
    def set_bounds(self, bounds):
        bounds = [int(b) for b in bounds]
        assert len(bounds) == self.world + 1 and bounds[0] == 0
        for g in range(self.world):
            assert bounds[g + 1] - bounds[g] >= MIN_SLAB_COLS, "slab not wider than the halo"
        self.bounds = bounds
        self.c0, self.c1 = bounds[self.rank], bounds[self.rank + 1]
        self.backend.set_owned(self.c0, self.c1)

    def set_caps(self, steady, first=None):
        """Message sizes (records) per slab boundary: `steady` between two rebalances, `first` for the next exchange
        only (the hand-over after the boundaries moved)."""
        self.caps = list(steady)
        self.caps_first = list(first) if first is not None else None
        reserve = getattr(self.backend, "reserve", None)
        if reserve is not None:
            reserve(max(list(steady) + (list(first) if first is not None else [])))
        self._ops_cache = {}

    def _caps_now(self):
        """(cap_left, cap_right) of the coming exchange; 0 where there is no neighbour."""
        c = self.caps_first if self.caps_first is not None else self.caps
        return (c[self.rank] if self.rank > 0 else 0), (c[self.rank + 1] if self.rank < self.world - 1 else 0)

    def rebalance(self):
        """Count-balanced boundaries from the global per-column histogram (one small all-reduce).  The same histogram
        re-sizes the halo messages, per boundary: every rank sees the same numbers, so sender and receiver stay in
        agreement.  The first exchange after the move hands over whole columns and gets a message sized for exactly
        that (handover_caps); from then on the steady size applies.  Returns the new boundaries."""
        # A rank whose asynchronous steps overflowed a buffer learns of it here (the histogram download synchronises and
        # raises).  It must still take part in the all-reduce, or its neighbours wait for it for ever: the failure
        # travels as one more element of the reduced vector and every rank raises.
        failure = None
        try:
            self.backend.drop_ghosts()
            local = self.backend.column_histogram().astype(np.int64)
        except Exception as e:  # noqa: BLE001 -- re-raised below, on every rank
            failure = e
            local = np.zeros(self.backend.drv.nx, np.int64)
        h = torch.from_numpy(np.concatenate([local, [1 if failure is not None else 0]]))
        dev = getattr(self.backend, "device", torch.device("cpu"))
        h = h.to(dev)
        dist.all_reduce(h, group=self.group)
        h = h.cpu().numpy()
        if h[-1]:
            if failure is not None:
                raise failure
            raise RuntimeError("slab rebalance: %d other rank(s) reported a failure (halo message or particle capacity "
                               "overflow); this rank's state is intact but the decomposition is not" % int(h[-1]))
        hist = h[:-1]
        self.last_hist = hist
        before = self.owned_counts()
        old = list(self.bounds)
        self.set_bounds(balanced_bounds(hist, self.world, prev=old))
        self.last_spread = (before, self.owned_counts())  # particles per rank under the old / the new boundaries
        steady = steady_caps(hist, self.bounds)
        self.set_caps(steady, handover_caps(hist, old, self.bounds, steady))
        self.rebalances += 1
        return self.bounds

    def owned_counts(self):
        """Particles per rank under the current boundaries, from the histogram of the last rebalance (None before)."""
        if self.last_hist is None:
            return None
        return [int(self.last_hist[self.bounds[g]:self.bounds[g + 1]].sum()) for g in range(self.world)]

    # -- halo exchange + migration ------------------------------------------------------------
    def _exchange_start(self, left, right, recv_l, recv_r):
        """One fixed-size message per direction (header + records); no count round trip.  Returns what
        _exchange_wait needs: the transfers are in flight (RCCL: on the communicator's own stream, ordered
        behind what the current stream has enqueued so far)."""
        r = self.rank
        stage = self.stage_cpu and (left if left is not None else right) is not None and \
            (left if left is not None else right).device.type != "cpu"
        out_l, out_r = recv_l, recv_r
        if stage:
            left = left.cpu() if left is not None else None
            right = right.cpu() if right is not None else None
            recv_l = torch.empty(recv_l.shape, dtype=recv_l.dtype) if left is not None else None
            recv_r = torch.empty(recv_r.shape, dtype=recv_r.dtype) if right is not None else None
        # (the backend's message buffers are fixed: the op list is built once and reused every step)
        key = tuple((t.data_ptr(), t.shape[0]) if t is not None else 0 for t in (left, right, recv_l, recv_r))
        ops = self._ops_cache.get(key) if not stage else None
        if ops is None:
            ops = []
            if left is not None:
                ops += [dist.P2POp(dist.isend, left, self._peer(r - 1), self.group),
                        dist.P2POp(dist.irecv, recv_l, self._peer(r - 1), self.group)]
            if right is not None:
                ops += [dist.P2POp(dist.isend, right, self._peer(r + 1), self.group),
                        dist.P2POp(dist.irecv, recv_r, self._peer(r + 1), self.group)]
            if not stage:
                self._ops_cache = {key: ops}
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, stage, (left, right, recv_l, recv_r, out_l, out_r)

    def _exchange_wait(self, handle):
        works, stage, (left, right, recv_l, recv_r, out_l, out_r) = handle
        for q in works:
            q.wait()  # RCCL: the current stream waits, the host does not
        if stage:
            if left is not None:
                out_l.copy_(recv_l)
            if right is not None:
                out_r.copy_(recv_r)

    def _exchange(self, left, right, recv_l, recv_r):
        self._exchange_wait(self._exchange_start(left, right, recv_l, recv_r))

    def _peer(self, group_rank):
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def run(self, nsteps, rebalance_every=0, mouse=(0.0, 0.0), strength=0.0):
        """nsteps steps with a rebalance every `rebalance_every` steps (0: never): what a long run of a scene whose
        mass moves (the collapsing dam) needs -- the slab loads diverge otherwise and the fullest columns outgrow the
        halo messages.  A rebalance synchronises the host (histogram download + all-reduce)."""
        done = 0
        while done < nsteps:
            k = nsteps - done if rebalance_every <= 0 else min(rebalance_every, nsteps - done)
            self.step(k, mouse, strength)
            done += k
            if rebalance_every > 0 and done < nsteps:
                self.rebalance()

    def step(self, nsteps=1, mouse=(0.0, 0.0), strength=0.0):
        """nsteps steps.  Before the first one the halo is exchanged on its own; between two steps of the
        same call the exchange overlaps the step: the backend first finishes the columns next to the slab
        boundaries (step_begin), the messages are packed from them and sent while the interior columns
        are computed (step_finish).  The state at return is that after a whole step, as if every step had
        been exchange, append, step.  (The last step of a call is deliberately not split: a split step hands its
        migrants on at once, as 20-byte records without density / pressure, and a download right after the call
        must still find every particle's density with its owner.)"""
        has_l, has_r = self.rank > 0, self.rank < self.world - 1
        send_hi_left = self.c0 + HALO_COLS if has_l else -1
        send_lo_right = self.c1 - HALO_COLS if has_r else BIG
        split = self.overlap and hasattr(self.backend, "step_begin") and (has_l or has_r)
        for i in range(nsteps):
            if i == 0 or not split:
                cl, cr = self._caps_now()
                left, right = self.backend.pack(send_hi_left, send_lo_right, cl, cr)
                recv_l, recv_r = self.backend.recv_buffers(cl, cr)
                self._exchange(left, right, recv_l, recv_r)
                self.backend.finish(cl, cr)
                self._sent(cl, cr)
            if not split or i == nsteps - 1:
                self.backend.step(mouse, strength)
                continue
            self.backend.step_begin(mouse, strength, has_l, has_r)
            cl, cr = self._caps_now()
            left, right = self.backend.pack(send_hi_left, send_lo_right, cl, cr)
            recv_l, recv_r = self.backend.recv_buffers(cl, cr)
            handle = self._exchange_start(left, right, recv_l, recv_r)
            self.backend.step_finish()
            self._exchange_wait(handle)
            self.backend.finish(cl, cr)
            self._sent(cl, cr)

    def _sent(self, cl, cr):
        self.records_sent += (cl + 1 if cl else 0) + (cr + 1 if cr else 0)
        self.exchanges += 1
        if self.caps_first is not None:  # the hand-over after a rebalance is done: back to the steady size
            self.caps_first = None
            self._ops_cache = {}

    def bytes_per_exchange(self):
        """Average bytes this rank has sent per exchange (both directions, headers included)."""
        return 20.0 * self.records_sent / max(self.exchanges, 1)

    # -- state --------------------------------------------------------------------------------
    def download_local(self):
        self.backend.drop_ghosts()
        return self.backend.download()

    def gather(self, n_total):
        """All particles in id order on every rank (verification / dumps; off the timed path)."""
        a, ids = self.download_local()
        parts = [None] * self.world
        dist.all_gather_object(parts, (a, ids), group=self.group)
        out = np.zeros(n_total, a.dtype)
        seen = np.zeros(n_total, np.int32)
        for pa, pid in parts:
            out[pid] = pa
            seen[pid] += 1
        if not np.all(seen == 1):
            raise RuntimeError("slab ownership broken: %d particles lost, %d duplicated" %
                               (int((seen == 0).sum()), int((seen > 1).sum())))
        return out

    def local_count(self):
        return self.backend.n

    def check(self):
        """Host synchronisation point: raises if any asynchronous step overflowed a buffer."""
        chk = getattr(self.backend, "check", None)
        return chk() if chk else self.backend.n

    def check_all(self):
        """check() as a collective: every rank calls it, and if any rank's check fails every rank raises (a rank that
        raised alone would leave the others waiting in their next collective).  Returns this rank's particle count."""
        failure, n = None, 0
        try:
            n = self.check()
        except Exception as e:  # noqa: BLE001 -- re-raised below
            failure = e
        flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int64, device=getattr(self.backend, "device", torch.device("cpu")))
        dist.all_reduce(flag, group=self.group)
        if failure is not None:
            raise failure
        if int(flag.item()):
            raise RuntimeError("slab check: %d other rank(s) reported a failure (halo message or particle capacity overflow)" % int(flag.item()))
        return n


def records_from_particles(a, ids):
    """28-byte particles + global ids -> 20-byte halo records."""
    r = np.zeros(len(a), HALO_DTYPE)
    r["pos"] = a["pos"]
    r["vel"] = a["vel"]
    r["idv"] = np.asarray(ids, np.uint32) | np.where(a["valid"] != 0, 0, 0x80000000).astype(np.uint32)
    return r


def distribute(slab, particles, ids=None):
    """Gives every rank the particles whose grid column it owns (each rank passes the same global
    array, or its own disjoint share with explicit ids whose owners are then fixed up by the
    normal migration of the first steps only if adjacent -- pass the full array when in doubt)."""
    ids = np.arange(len(particles)) if ids is None else np.asarray(ids)
    col = grid_column(particles["pos"][:, 0], slab.params, slab.backend.drv)
    mine = (col >= slab.c0) & (col < slab.c1)
    slab.backend.append_host(records_from_particles(particles[mine], ids[mine]))
    return int(mine.sum())


def make_hip_slab(params, particles_or_fn, device, group=None, hist=None):
    """Builds the HIP-backed SlabSolver.  `particles_or_fn` is the global particle array, or a
    function (c0, c1) -> (particles, ids) generating only the owned share (large scenes)."""
    drv = _lib.derive(params)
    world = dist.get_world_size(group)
    if hist is None:
        col = grid_column(particles_or_fn["pos"][:, 0], params, drv)
        hist = np.bincount(col, minlength=drv.nx)
    bounds = balanced_bounds(hist, world)
    per = int(np.max([np.sum(hist[bounds[g]:bounds[g + 1]]) for g in range(world)]))
    halo_cap = max(steady_caps(hist, bounds) + [HALO_FLOOR])
    # particle capacity: the balanced share with room for imbalance between two rebalances, plus ghosts (two halos,
    # last step's stale copies and the fresh ones) with room to grow
    n_max = int(per * 1.5) + 8 * halo_cap + 1024
    backend = HipSlabBackend(params, n_max, device, halo_cap)
    slab = SlabSolver(backend, params, bounds, group, hist=hist)
    if callable(particles_or_fn):
        a, ids = particles_or_fn(slab.c0, slab.c1, drv)
        backend.append_host(records_from_particles(a, ids))
    else:
        distribute(slab, particles_or_fn)
    return slab
