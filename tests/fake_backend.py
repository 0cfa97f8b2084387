"""CPU stand-in for HipSlabBackend, TEST INFRASTRUCTURE ONLY: same interface, numpy/torch-CPU
buffers, the physics done by the oracle (oracle/sph_oracle.c).  It lets the slab logic of
sph_amd/slab.py (partition, packing rules, exchange, migration, ghost dropping, rebalancing) run
under gloo with world_size > 1 on a machine without a GPU.  Ghosts pass through a step unchanged,
as in the library (k_force pass-through for non-owned columns)."""
import numpy as np
import torch

import orc
from sph_amd import lib as _lib
from sph_amd.params import HALO_DTYPE, PARTICLE_DTYPE
from sph_amd.slab import HALO_COLS, grid_column


class FakeSlabBackend:
    def __init__(self, params):
        self.params = params
        self.drv = _lib.derive(params)  # host-only call, no GPU needed
        self.O = orc.Oracle()
        self.a = np.zeros(0, PARTICLE_DTYPE)
        self.ids = np.zeros(0, np.int64)
        self.owned_mask = None  # set by step(): which entries were owned when the step sorted them
        self.device = torch.device("cpu")
        self.lo, self.hi = 0, self.drv.nx

    CAP = 4096  # message size when the SlabSolver was given no histogram to size them from

    def recv_buffers(self, cap_left, cap_right):
        self.recv_l = torch.zeros((cap_left + 1, 5), dtype=torch.int32) if cap_left else None
        self.recv_r = torch.zeros((cap_right + 1, 5), dtype=torch.int32) if cap_right else None
        return self.recv_l, self.recv_r

    def _message(self, rec, cap):
        # the library flags an overflow (sticky error, records dropped); here it fails the test at once
        assert rec.shape[0] <= cap, "halo message overflow: %d records, capacity %d" % (rec.shape[0], cap)
        self.max_fill = max(getattr(self, "max_fill", 0.0), rec.shape[0] / cap)
        m = torch.zeros((cap + 1, 5), dtype=torch.int32)
        m[0, 0] = rec.shape[0]
        m[1:1 + rec.shape[0]] = rec
        return m

    def finish(self, cap_left, cap_right):
        if cap_left:
            self.append(self.recv_l[1:1 + int(self.recv_l[0, 0])])
        if cap_right:
            self.append(self.recv_r[1:1 + int(self.recv_r[0, 0])])

    def set_owned(self, lo, hi):
        self.lo, self.hi = lo, hi

    def _records(self, mask):
        r = np.zeros(int(mask.sum()), HALO_DTYPE)
        r["pos"] = self.a["pos"][mask]
        r["vel"] = self.a["vel"][mask]
        r["idv"] = self.ids[mask].astype(np.uint32) | np.where(self.a["valid"][mask] != 0, 0, 0x80000000).astype(np.uint32)
        return torch.from_numpy(r.view(np.int32).reshape(-1, 5).copy())

    def drop_ghosts(self):
        if self.owned_mask is not None:
            self.a, self.ids = self.a[self.owned_mask], self.ids[self.owned_mask]
            self.owned_mask = None

    def pack(self, send_hi_left, send_lo_right, cap_left, cap_right):
        self.drop_ghosts()
        col = grid_column(self.a["pos"][:, 0], self.params, self.drv)
        left = self._message(self._records(col < send_hi_left), cap_left) if cap_left else None
        right = self._message(self._records(col >= send_lo_right), cap_right) if cap_right else None
        keep = (col >= self.lo - HALO_COLS) & (col < self.hi + HALO_COLS)  # fresh migrants stay as ghosts
        self.a, self.ids = self.a[keep], self.ids[keep]
        return left, right

    def append(self, rec):
        if rec.shape[0] == 0:
            return
        self.drop_ghosts()
        r = np.ascontiguousarray(rec.numpy()).view(np.uint8).reshape(-1).view(HALO_DTYPE)
        b = np.zeros(len(r), PARTICLE_DTYPE)
        b["pos"], b["vel"] = r["pos"], r["vel"]
        b["density"] = self.params.rest_density
        b["valid"] = (r["idv"] & 0x80000000) == 0
        cat = np.zeros(len(self.a) + len(b), PARTICLE_DTYPE)  # np.concatenate would drop the 28-byte padding
        cat[: len(self.a)] = self.a
        cat[len(self.a):] = b
        self.a = cat
        self.ids = np.concatenate([self.ids, (r["idv"] & 0x7FFFFFFF).astype(np.int64)])

    def append_host(self, rec_np):
        self.append(torch.from_numpy(np.ascontiguousarray(rec_np).view(np.int32).reshape(-1, 5).copy()))

    def step(self, mouse, strength):
        # canonical order is (key, id): present the particles to the oracle sorted by id so that its
        # in-cell tie-break (index) equals the id order the GPU uses
        o = np.argsort(self.ids, kind="stable")
        self.a, self.ids = self.a[o], self.ids[o]
        col = grid_column(self.a["pos"][:, 0], self.params, self.drv)
        owned = (col >= self.lo) & (col < self.hi)
        before = self.a.copy()
        self.O.step_cells(self.params, self.a, 1, mouse, strength)
        # ghosts pass through (their results lack neighbours beyond the halo)
        for f in ("pos", "vel"):
            self.a[f][~owned] = before[f][~owned]
        self.owned_mask = owned

    def step_begin(self, mouse, strength, has_left, has_right):
        self.step(mouse, strength)  # (the oracle has no halves: everything is finished before the pack)

    def step_finish(self):
        pass

    def download(self):
        o = np.argsort(self.ids, kind="stable")
        return self.a[o].copy(), self.ids[o].astype(np.int32)

    def column_histogram(self):
        return np.bincount(grid_column(self.a["pos"][:, 0], self.params, self.drv), minlength=self.drv.nx).astype(np.int32)

    @property
    def n(self):
        return len(self.a)
