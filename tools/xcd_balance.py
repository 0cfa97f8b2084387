"""How evenly the pair kernels' work list loads the 8 XCDs: the list is cut into 8 contiguous chunks of equal LENGTH
(for_each_tile, pairs.hip), but entries differ in cost once the scene has evolved.  Rebuilds the base tiles on the host
from the cell table (column-major, `rows` cells high, empty ones skipped), prices each by own particles x stencil
candidates (what both pair kernels loop over), and prints the cost share of the fullest chunk for the contiguous split
and for block-cyclic splits.  python tools/xcd_balance.py [evolve_steps] [cols]"""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

evolve = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
params, a = scenes.dam_break(cols, cols)
with Solver(params, n_max=len(a)) as s:
    s.upload(a)
    done = 0
    for upto in sorted({0, evolve // 2, evolve}):
        if upto:
            s.stepSimulation(nsteps=upto - done)
        done = upto
        s.buildNeighbors()
        _, _, cs, ce = s.cells()
        rows = s.tile_stats()["tile_rows"] or 15
        nx, ny = s.drv.nx, s.drv.ny
        cnt = (ce - cs).astype(np.int64).reshape(nx, ny)
        cand = np.zeros_like(cnt)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                sx0, sx1 = max(0, dx), nx + min(0, dx)
                sy0, sy1 = max(0, dy), ny + min(0, dy)
                cand[sx0 - dx:sx1 - dx, sy0 - dy:sy1 - dy] += cnt[sx0:sx1, sy0:sy1]
        blocks = -(-ny // rows)
        pad = blocks * rows - ny
        own = np.pad(cnt, ((0, 0), (0, pad))).reshape(nx, blocks, rows).sum(axis=2)
        cost = np.pad(cnt * cand, ((0, 0), (0, pad))).reshape(nx, blocks, rows).sum(axis=2)
        keep = own.reshape(-1) > 0  # list order: column-major
        cost = cost.reshape(-1)[keep].astype(np.float64)
        own = own.reshape(-1)[keep]
        total = len(cost)
        per = -(-total // 8)
        shares = [cost[k * per:(k + 1) * per].sum() for k in range(8)]
        line = "step %4d: %d base tiles of %d rows, own/tile mean %.0f max %d; fullest XCD / mean: contiguous %.3f" % (
            upto, total, rows, own.mean(), own.max(), max(shares) / (sum(shares) / 8))
        for C in (8, 16, 32, 64, 128, 256):
            xcd = (np.arange(total) // C) % 8
            sh = np.bincount(xcd, weights=cost, minlength=8)
            line += ", cyclic %d: %.3f" % (C, sh.max() / sh.mean())
        print(line, flush=True)
