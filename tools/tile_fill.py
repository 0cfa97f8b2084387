"""Lane fill of the pair kernels' work list, emulated on the host from the cell populations of a scene:
the current scheme (base tiles of T rows, cut where they do not fit) against greedy packing down each column.
    python tools/tile_fill.py [cols=1024] [evolve_steps=1000]"""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from sph_amd import scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
evolve = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
CAP, LANES, PAD, MAXR = 960, 256, 3, 16


def pack(cnt, col, a_lo, a_hi):
    """greedy tiles over rows [a_lo, a_hi) of column col -> list of (rows, own)"""
    nx, ny = cnt.shape
    out = []
    a = a_lo
    c3 = cnt[max(col - 1, 0):col + 2].sum(axis=0)
    while a < a_hi:
        if cnt[col, a] == 0:
            a += 1
            continue
        own, b = 0, a
        while b < a_hi and b - a < MAXR:
            o2 = own + cnt[col, b]
            cand = c3[max(a - 1, 0):min(b + 2, ny)].sum() + PAD * (min(b + 2, ny) - max(a - 1, 0))
            if b > a and (o2 > LANES or cand > CAP):
                break
            own = o2
            b += 1
        while b - 1 > a and cnt[col, b - 1] == 0:  # trailing empty rows
            b -= 1
        out.append((b - a, int(own)))
        a = b
    return out


def report(name, tiles, n):
    own = np.array([t[1] for t in tiles])
    wg = len(own)
    waves = np.ceil(own / 64.0).sum()
    print("  %-28s workgroups %6d  fill of 256 lanes %.3f  fill of occupied waves %.3f  (own>256: %d)" % (
        name, wg, n / (256.0 * wg), n / (64.0 * waves), int((own > LANES).sum())), flush=True)


for steps in (0, evolve):
    p, a = scenes.dam_break(cols)
    with Solver(p, n_max=len(a)) as s:
        s.upload(a)
        if steps:
            s.stepSimulation(nsteps=steps)
        s.buildNeighbors()
        keys, order, cs, ce = s.cells()
        nx, ny = s.drv.nx, s.drv.ny
    cnt = (ce - cs).reshape(nx, ny)
    n = int(cnt.sum())
    print("dam %d x %d after %d steps: grid %d x %d, per non-empty cell mean %.1f max %d" % (cols, cols, steps, nx, ny, cnt[cnt > 0].mean(), cnt.max()))
    for T in (12, 13, 14, 15, 16):
        tiles = []
        for col in np.nonzero(cnt.sum(axis=1))[0]:
            for a0 in range(0, ny, T):
                tiles += pack(cnt, col, a0, min(a0 + T, ny))
        report("base tiles of %d rows + cuts" % T, tiles, n)
    tiles = []
    for col in np.nonzero(cnt.sum(axis=1))[0]:
        tiles += pack(cnt, col, 0, ny)
    report("greedy packing per column", tiles, n)
