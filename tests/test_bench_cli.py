"""bench.py's command line on a box without a GPU: the multi-GPU line must not depend on an external launcher any
more (round 2: `python bench.py --gpus 8` raised SystemExit unless torch.distributed.run had started it), and every
refusal is a clear message with a non-zero exit code -- never a hang, never a silent CPU fallback."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

needs_no_gpu = pytest.mark.skipif(torch.cuda.device_count() > 0, reason="checks the refusals of a GPU-less box")


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=e)


@needs_no_gpu
def test_single_gpu_bench_refuses_without_a_gpu():
    r = _run(["--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and "needs a GPU" in r.stderr and "no CPU fallback" in r.stderr


@needs_no_gpu
@pytest.mark.parametrize("extra", [[], ["--driver", "mg"]])
def test_multi_gpu_bench_without_launcher_refuses_cleanly(extra):
    """No WORLD_SIZE: bench.py would start its own ranks (or the one-process driver) -- here it must say that there is
    no GPU, before spawning anything."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"] + extra)
    assert r.returncode != 0 and "needs a GPU" in r.stderr
    assert "torch.distributed.run with --nproc-per-node" not in r.stderr  # (round 2's message)


def test_launcher_mismatch_and_bad_arguments():
    r = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr
    r = _run(["--gpus", "2", "--driver", "mg"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "ONE process" in r.stderr
    assert _run(["--gpus", "0"]).returncode != 0
    assert _run(["--workload", "nope"]).returncode != 0


def test_workloads_are_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.WORKLOADS["dam1m"] == ("dam", 1024, 1024)      # configs[1]
    assert bench.WORKLOADS["box4m"] == ("box", 2048, 2048)      # configs[2]
    assert bench.WORKLOADS["dam16m"] == ("dam", 4096, 4096)     # configs[3]
    assert bench.WORKLOADS["dam8m"][1] ** 2 > 8_000_000         # configs[4]: 8M per GPU
    s = bench.stats([5.0e9, 5.1e9, 4.9e9])
    assert s["repeats"] == 3 and s["value_median"] == 5.0e9 and abs(s["spread"] - 0.04) < 1e-12


def test_zero_pressure_tile_share_on_a_small_grid():
    """bench.py's host recount of which base tiles stage only +0 pressures (the force_terms<ZEROP> path)."""
    import numpy as np

    sys.path.insert(0, ROOT)
    import bench

    nx, ny = 4, 6  # column-major cell keys; one particle per cell, column 3 and cell (0, 5) empty
    counts = np.ones((nx, ny), int)
    counts[3, :] = 0
    counts[0, 5] = 0
    cs = np.concatenate([[0], np.cumsum(counts.reshape(-1))])[:-1]
    ce = cs + counts.reshape(-1)
    n = int(counts.sum())
    order = np.random.RandomState(1).permutation(n)  # particle id per sorted slot
    bits = np.zeros(n, np.uint32)
    assert bench.zero_pressure_tiles(cs, ce, order, bits, nx, ny, 2) == (9, 9)
    bits[order[cs[1 * ny + 2]]] = 0x3F800000  # the particle of cell (1, 2) has P = 1
    # its pressure is staged by every tile that owns a cell of columns 0..2, rows 1..3
    assert bench.zero_pressure_tiles(cs, ce, order, bits, nx, ny, 2) == (9, 3)  # the row blocks {4, 5} stay clear
    assert bench.zero_pressure_tiles(cs, ce, order, bits, nx, ny, 3) == (6, 0)
    bits[:] = 0
    bits[order[cs[2 * ny + 5]]] = 0x80000000  # -0 is not +0: the kernel compares bits
    assert bench.zero_pressure_tiles(cs, ce, order, bits, nx, ny, 3) == (6, 4)  # hot: rows 3..5 of columns 1 and 2
