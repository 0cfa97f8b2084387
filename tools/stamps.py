#!/usr/bin/env python3
"""Phase timing inside k_force from the -DSPH_STAMPS diagnostic build (tools/build_variant.sh stamps "-DSPH_STAMPS"):
   SPHHIP_LIB=$PWD/variants/stamps/libsphhip.so python tools/stamps.py [cols] [steps]
Every wave stamps s_memtime at: 0 tile start, 1 tile_setup done, 2 candidates staged, 3 pair loop done, 4 epilogue
done.  Prints the distribution of the phase lengths (shader cycles) and the timeline of the launch."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sph_amd import lib as _lib, scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6  # (e.g. 1000: the evolved scene)
params, a = scenes.dam_break(cols)
L = _lib.load()
L.sphhip_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
with Solver(params, n_max=len(a)) as s:
    s.upload(a)
    s.stepSimulation(nsteps=steps)
    s.synchronize()
    tiles = 1 << 16
    buf = np.zeros((tiles, 4, 8), np.uint64)
    assert L.sphhip_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), tiles) == 0
used = buf[:, 0, 0] != 0
b = buf[used].astype(np.int64)
print("tiles stamped:", used.sum())
own = b[:, 0, 5] & 0xFFFFFFFF
tot = b[:, 0, 5] >> 32
print("own particles per tile: mean %.1f max %d; candidates: mean %.1f max %d" % (own.mean(), own.max(), tot.mean(), tot.max()))
names = ["table", "stage", "loop", "epilogue", "whole"]
for w in range(4):
    t = b[:, w, :5]
    ok = (t[:, 0] != 0) & (t[:, 4] != 0)
    d = np.stack([t[ok, 1] - t[ok, 0], t[ok, 2] - t[ok, 1], t[ok, 3] - t[ok, 2], t[ok, 4] - t[ok, 3], t[ok, 4] - t[ok, 0]], 1)
    print("wave %d (%d tiles):" % (w, ok.sum()), "  ".join("%s p50 %d p90 %d mean %.0f" % (n, np.percentile(d[:, i], 50), np.percentile(d[:, i], 90), d[:, i].mean()) for i, n in enumerate(names)))
# timeline in real time (100 MHz ticks) of tile starts; shader clock = s_memtime cycles per s_memrealtime tick
rt = b[:, 0, 7]
rt0 = rt.min()
start_us = (rt - rt0) / 100.0
end_us = (b[:, :, 6].max(1) - rt0) / 100.0
whole = (b[:, 0, 4] - b[:, 0, 0])
real = (b[:, 0, 6] - b[:, 0, 7])
ok = real > 0
print("shader clock (s_memtime cycles per us of s_memrealtime): median %.0f MHz" % np.median(whole[ok] / (real[ok] / 100.0)))
print("launch: last tile start at %.1f us, last end at %.1f us" % (start_us.max(), end_us.max()))
hist, edges = np.histogram(start_us, bins=12)
print("tile starts per %.1f-us bin:" % (edges[1] - edges[0]), hist.tolist())
print("mean tile length %.0f cycles = %.1f us, p90 %.0f" % (whole.mean(), (real[ok] / 100.0).mean(), np.percentile(whole, 90)))
# concurrency over time: tiles in flight per 5-us bin (1024 = every workgroup slot of the GPU busy at 4 per CU)
edges = np.arange(0.0, end_us.max() + 5.0, 5.0)
infl = [(int(((start_us < e1) & (end_us > e0)).sum())) for e0, e1 in zip(edges[:-1], edges[1:])]
print("tiles in flight per 5-us bin:", infl)
busy = np.clip(np.minimum(end_us[:, None], edges[None, 1:]) - np.maximum(start_us[:, None], edges[None, :-1]), 0, None).sum(0) / 5.0
print("mean tiles resident per 5-us bin:", [int(x) for x in busy])
print("sum of tile lengths / 1024 slots = %.1f us" % ((end_us - start_us).sum() / 1024.0))
# per XCD (contiguous eighths of the list): tiles, sum of tile lengths, time of the last end
idx = np.nonzero(used)[0]
per = (idx.max() + 8) // 8
for x in range(8):
    m = (idx // per) == x
    print("XCD %d: %4d tiles, own particles %7d, sum of tile lengths %7.0f us, first start %5.1f, last end %6.1f us" % (
        x, m.sum(), own[m].sum(), (end_us[m] - start_us[m]).sum(), start_us[m].min(), end_us[m].max()))
