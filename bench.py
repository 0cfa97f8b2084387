#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the full SPH step (cell list + density/pressure +
force + integrate/boundary) with the state resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dam1m|box4m|dam16m|dam8m|...]
                  [--scaling weak|strong] [--driver c10d|mg] [--repeats R]

N = 1: BASELINE.json configs[1], the 2-D dam break with 1M particles (1024 x 1024 lattice at
spacing H/4 against the left wall of a 4W x 2W box, all sph.h constants).

N > 1: BASELINE.json configs[3], the 16M-particle dam break split into N x-slabs (strong scaling;
ghost-column halo exchange and particle migration between x-neighbours over RCCL), and in the same
JSON line configs[4] (8M particles per GPU, weak scaling) and the 16M scene on one GPU.  One rank
per GPU over torch.distributed (sph_amd/slab.py): launched by `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`, or by this script itself when it is started plainly
(`python bench.py --gpus N`): the ranks are then spawned as fresh child processes before this
process has touched the GPU.  `--driver mg` runs the same decomposition through the C++ driver
libsph_mg.so instead: ONE process, ncclCommInitAll, one enqueue thread per slab
(include/sph_mg.h).  On a box with fewer GPUs than N the run is refused unless
SPH_BENCH_SHARE_GPU=1 puts all slabs on device 0 (a protocol rehearsal, not a measurement:
messages then travel through host memory / device copies).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (fused force + integrate)
by its ALGORITHMIC bytes (40 B/particle: R pos 8, vel 8, rho 4, P 4; W pos 8, vel 8; neighbour
gathers are served by LDS/L2 and not counted) over its average launch time,
measured with HIP events on the stream the kernel runs on, inside the timed region.
`cpu_baseline` is the oracle's portable restatement of the reference's OpenMP path
(SPH/openMP_sph.cpp:80-205, all-pairs) timed on this host's cores on a bounded miniature of the
same scene.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X peak FP32 vector (MI355X_MICROARCH.md)
# HIP events around k_force in one step of this many inside the timed region: an event pair leaves ~11 us of gaps on the
# stream (rocprofv3 kernel trace), i.e. bracketing every step would slow the measured loop itself by 4-5 %
FORCE_SAMPLE_PERIOD = 8
FORCE_BYTES_PER_PARTICLE = 40  # fused force + integrate, SURVEY.md 8(d)
# algorithmic flops, SURVEY.md 8(d): per stencil candidate 7 (2 sub, 2 mul, add, cmp, select), per in-range density
# pair +6, per in-range force pair +45 (div and sqrt counted as 1 each), per particle epilogue ~25
FLOP_CANDIDATE, FLOP_DENSITY_PAIR, FLOP_FORCE_PAIR, FLOP_EPILOGUE = 7, 6, 45, 25
PAIR_SLOTS_PER_CU = 4  # workgroups of k_force resident per CU (LDS budget of 960 records, pairs.hip)

WORKLOADS = {
    # name: (kind, lattice columns (per GPU when weak scaling), rows)
    "dam1m": ("dam", 1024, 1024),
    "dam4m": ("dam", 2048, 2048),
    "dam16m": ("dam", 4096, 4096),
    "dam8m": ("dam", 2896, 2896),
    "box4m": ("box", 2048, 2048),
    "dam64k": ("dam", 256, 256),
    "dam256k": ("dam", 512, 512),
}


def cpu_baseline():
    """Times the oracle's OpenMP all-pairs port (and the cell-list oracle) on this host."""
    import numpy as np  # noqa: F401

    from sph_amd import scenes

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import orc

        O = orc.Oracle()
    except Exception as e:  # oracle not built: report, do not fail the bench
        return {"value": None, "unit": "particle-updates/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}, {}
    cores = O.num_threads()
    cols = 256 if cores < 12 else 362
    params, a = scenes.dam_break(cols)
    O.omp_step_allpairs(params, a, 0)
    steps = 8  # ~10 s of CPU work on 16 cores
    t0 = time.perf_counter()
    O.omp_step_allpairs(params, a, steps)
    dt = time.perf_counter() - t0
    base = {
        "value": len(a) * steps / dt,
        "unit": "particle-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": "OpenMP all-pairs restatement of openMP_sph.cpp:80-205 on a %d-particle dam-break miniature "
        "(same spacing H/4, %d x %d lattice), %d steps, %.1f s; O(N^2): at 1M particles it would be %.0fx slower "
        "per particle" % (len(a), cols, cols, steps, dt, 1048576.0 / len(a)),
    }
    extra = {}
    params, a = scenes.dam_break(1024)
    t0 = time.perf_counter()
    O.step_cells(params, a, 1)
    dt = time.perf_counter() - t0
    extra["cpu_celllist"] = {
        "value": len(a) / dt, "unit": "particle-updates/s", "cores": cores, "kind": "port",
        "sample": "oracle cell-list step (same uniform grid as the HIP path, OpenMP) on the full 1M dam-break, 1 step, %.1f s" % dt,
    }
    if orc.have_ref():
        R = orc.Reference()
        b = R.init(8192)
        t0 = time.perf_counter()
        R.step(b, 2)
        dt = time.perf_counter() - t0
        extra["cpu_reference_serial"] = {
            "value": 8192 * 2 / dt, "unit": "particle-updates/s", "cores": 1, "kind": "reference",
            "sample": "reference SPH/sph.cpp compiled unmodified (oracle/_ref), its own lattice scene N=8192, 2 steps, %.1f s" % dt,
        }
    # the reference's own harness and unit (X_performance_test: steps per second on its default scene, N = 1024,
    # SPH/serial_main.cpp:120-147 / main.cpp:115,127) through the oracle-side dispatcher, shortened to 2 runs x 50 steps
    exe = os.path.join(ROOT, "oracle", "sph_cpu_main")
    if os.path.exists(exe):
        ups = {}
        for method, label in (("serial", "Serial"), ("omp", "OpenMP")):
            try:
                r = subprocess.run([exe, "--method", method, "--n", "1024", "--runs", "2", "--steps", "50"], capture_output=True, text=True, timeout=120)
                for line in r.stdout.splitlines():
                    if line.startswith(label + " Average UPS:"):
                        ups[label + " Average UPS"] = float(line.split(":")[1])
            except Exception:
                pass
        if ups:
            ups["sample"] = "oracle/sph_cpu_main --n 1024 --runs 2 --steps 50 (reference default scene, its UPS unit; OpenMP on %d threads)" % cores
            extra["cpu_reference_menu"] = ups
    return base, extra


def stencil_candidates(solver, drv):
    """Sum over particles of the population of their 3x3 cell stencil (the candidates a cell-list pass tests), from
    the cell table of the current grid.  Host numpy, off the timed path."""
    import numpy as np

    solver.buildNeighbors()
    _, _, cs, ce = solver.cells()
    cnt = (ce - cs).astype(np.int64).reshape(drv.nx, drv.ny)
    acc = np.zeros_like(cnt)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            sx0, sx1 = max(0, dx), drv.nx + min(0, dx)
            sy0, sy1 = max(0, dy), drv.ny + min(0, dy)
            acc[sx0 - dx:sx1 - dx, sy0 - dy:sy1 - dy] += cnt[sx0:sx1, sy0:sy1]
    return int((cnt * acc).sum())


def zero_pressure_tiles(cell_start, cell_end, order, pressure_bits, nx, ny, tile_rows):
    """Share of k_force's base tiles (one grid column x `tile_rows` rows with at least one particle) whose staged
    candidates -- the 3x3 stencils of their cells -- all have pressure bits +0: those run the instantiation without the
    pressure term (pairs.hip, force_terms<ZEROP>).  Host numpy over the cell table, off the timed path.  Tiles the device
    cuts into sub-tiles stage smaller stencils, so the true share is at least this one.  -> (tiles, zero-pressure tiles)"""
    import numpy as np

    nz_sorted = (np.asarray(pressure_bits)[np.asarray(order)] != 0).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(nz_sorted)])
    cs, ce = np.asarray(cell_start, np.int64), np.asarray(cell_end, np.int64)
    cell_nz = (csum[ce] - csum[cs]).reshape(nx, ny) > 0
    cell_used = (ce - cs).reshape(nx, ny) > 0
    near = np.zeros_like(cell_nz)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            sx0, sx1 = max(0, dx), nx + min(0, dx)
            sy0, sy1 = max(0, dy), ny + min(0, dy)
            near[sx0 - dx:sx1 - dx, sy0 - dy:sy1 - dy] |= cell_nz[sx0:sx1, sy0:sy1]
    blocks = -(-ny // tile_rows)
    pad = blocks * tile_rows - ny
    used = np.pad(cell_used, ((0, 0), (0, pad))).reshape(nx, blocks, tile_rows).any(axis=2)
    # (a tile stages the stencils of ALL its rows, empty cells included)
    hot = np.pad(near, ((0, 0), (0, pad))).reshape(nx, blocks, tile_rows).any(axis=2)
    return int(used.sum()), int((used & ~hot).sum())


def pressure_paths(solver, params):
    """Which force_terms instantiation the scene's tiles run: needs the densities / pressures of the current state
    (call after computeDensityPressure)."""
    import numpy as np

    state = solver.download()
    bits = np.ascontiguousarray(state["pressure"]).view(np.uint32)
    nz = int(np.count_nonzero(bits))
    out = {"particles_with_nonzero_pressure": nz, "max_density": float(state["density"].max()) if len(state) else 0.0,
           "rest_density": float(params.rest_density)}
    if nz == 0:
        out["zero_pressure_tile_frac"] = 1.0
        out["note"] = "every pressure is +0 (P = max(0, K (rho - rho0)), sph.cpp:52-53): every tile runs force_terms<ZEROP>"
    else:
        _, order, cs, ce = solver.cells()
        rows = solver.tile_stats()["tile_rows"]
        tiles, zero = zero_pressure_tiles(cs, ce, order, bits, solver.drv.nx, solver.drv.ny, max(rows, 1))
        out["zero_pressure_tile_frac"] = zero / max(tiles, 1)
        out["base_tiles"] = tiles
        out["note"] = "base tiles (column x %d rows) whose 3x3 stencils hold no particle with P != +0; a lower bound where the device cuts tiles" % rows
    return out


def force_rooflines(n_launch, force_avg_s, pairs_local, cand_local, traffic=None, valu=None, kernel="k_force<fused integrate>"):
    """The two roofline blocks of the dominant kernel (fused force + integrate): HBM by algorithmic bytes, and FP32
    vector by SURVEY.md 8(d)'s algorithmic flop counts (the bound that actually applies at ~50 neighbours).
    n_launch: particles the timed launch updates."""
    achieved = FORCE_BYTES_PER_PARTICLE * n_launch / force_avg_s / 1e9 if force_avg_s > 0 else 0.0
    hbm = {
        "kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": force_avg_s * 1e3,
        "algorithmic_bytes_per_launch": FORCE_BYTES_PER_PARTICLE * n_launch, "valu": valu,
        "note": "pair kernels are VALU-bound at ~50 neighbours (no MFMA: irregular gathers); see roofline_valu and DESIGN.md",
    }
    fp32 = None
    if cand_local:
        flops = FLOP_CANDIDATE * cand_local + FLOP_FORCE_PAIR * pairs_local + FLOP_EPILOGUE * n_launch
        tf = flops / force_avg_s / 1e12 if force_avg_s > 0 else 0.0
        fp32 = {
            "kernel": kernel, "bound": "valu_fp32", "achieved": tf, "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TFLOPS, "flops_per_launch": flops,
            "model": "SURVEY.md 8(d): 7 per stencil candidate + 45 per in-range pair + 25 per particle; candidates %d, in-range "
                     "pairs %d" % (cand_local, pairs_local),
        }
    return hbm, fp32


def committed_pmc(name):
    """HBM bytes per k_force launch and the VALU issue figures of the committed PMC profile of this workload
    (profiles/traffic.json names it); None where there is none."""
    traffic, valu = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        ent = json.load(open(tf)).get(name, {})
        traffic = ent.get("k_force_bytes_per_launch")
        pm = json.load(open(os.path.join(ROOT, ent["source"])))
        kf = [v for k, v in pm.items() if k.startswith("k_force<true")][0]
        valu = {"cycles_per_wave_instruction": kf.get("valu_cycles_per_wave_instr"), "lane_utilisation": kf.get("valu_lane_utilisation"),
                "wave_instructions_per_launch": kf.get("SQ_INSTS_VALU"), "source": ent["source"]}
    except Exception:
        pass
    return traffic, valu


def stats(values):
    import numpy as np

    v = np.asarray(values, float)
    return {"repeats": len(v), "value_median": float(np.median(v)), "value_min": float(v.min()), "value_max": float(v.max()),
            "spread": float((v.max() - v.min()) / np.median(v)) if len(v) else 0.0}


def tile_rounds(solver):
    """How the pair kernels' work list fills the GPU: entries, resident workgroup slots, rounds, and the share of the
    last round's slots that stay idle (the launch tail)."""
    st = solver.tile_stats()
    slots = PAIR_SLOTS_PER_CU * 256
    tiles = st["tiles"]
    if tiles <= 0:
        return None
    rounds = tiles / slots
    full = -(-tiles // slots)
    return {"tiles": tiles, "slots": slots, "rounds": rounds, "tail_frac": (full * slots - tiles) / float(full * slots),
            "long": st["long"], "dense_force": st["dense_force"]}


def run_single(name, steps, warmup, device, evolve=0, repeats=1, prime=0):
    """One single-GPU scene: `evolve` untimed steps, `warmup` untimed steps, `repeats` timed intervals of `steps` steps
    with HIP events around the force kernel, then neighbour statistics.  `value` is the FIRST interval."""
    import torch

    from sph_amd import scenes
    from sph_amd.solver import Solver

    kind, cols, rows = WORKLOADS[name]
    params, a = scenes.dam_break(cols, rows) if kind == "dam" else scenes.uniform_box(cols)
    n = len(a)
    with Solver(params, n_max=n, device=device) as solver:
        solver.upload(a)
        if prime:
            # Set-up, not part of the measured scene: `prime` steps, then the initial state is uploaded again.  A GPU that
            # has idled through the scene generation and the upload needs tens of milliseconds under load to reach its
            # clocks, and the library's launch sizing follows what its kernels report back; with only `warmup` = 5 steps
            # (1 ms) in front of it the first timed interval came out 4-7 % below the following ones.
            solver.stepSimulation(nsteps=prime)
            solver.upload(a)
        del a
        if evolve:
            solver.stepSimulation(nsteps=evolve)
        solver.stepSimulation(nsteps=warmup)
        torch.cuda.synchronize()
        solver.set_profiling_phases(["force"])
        solver.set_profiling_period(FORCE_SAMPLE_PERIOD)
        solver.reset_phase_times()
        times = []
        for _ in range(max(repeats, 1)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solver.stepSimulation(nsteps=steps)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        elapsed = times[0]
        f_ms, f_k = solver.phase_times()["force"]
        solver.set_profiling(False)
        rounds = tile_rounds(solver)
        # per-phase breakdown: a few extra steps with every phase bracketed (not part of `value`)
        solver.set_profiling_period(1)
        solver.set_profiling(True)
        solver.reset_phase_times()
        solver.stepSimulation(nsteps=5)
        torch.cuda.synchronize()
        breakdown = {k: v[0] / 5 for k, v in solver.phase_times().items() if v[1]}
        solver.set_profiling(False)
        solver.computeDensityPressure()
        pairs = solver.pair_count()
        counts = solver.neighbor_counts()
        cand = stencil_candidates(solver, solver.drv)
        paths = pressure_paths(solver, params)
        force_avg_s = (f_ms / max(f_k, 1)) * 1e-3  # (f_k: the sampled launches)
        traffic, valu = committed_pmc(name if not evolve else "-")
        hbm, fp32 = force_rooflines(n, force_avg_s, pairs, cand, traffic, valu)
        out = {
            "workload": name, "particles": n, "grid": [solver.drv.nx, solver.drv.ny], "steps": steps, "warmup": warmup,
            "evolved_steps": evolve, "value": n * steps / elapsed, "unit": "particle-updates/s",
            "ms_per_step": elapsed / steps * 1e3, "mean_neighbors": pairs / max(n, 1),
            "max_neighbors": int(counts.max()) - 1 if n else 0,
            "pair_interactions_per_s": 2.0 * pairs * steps / elapsed,
            "roofline": hbm, "roofline_valu": fp32, "phases_ms_per_step": breakdown, "tile_rounds": rounds,
            "pressure_paths": paths, "box": [float(params.xmax - params.xmin), float(params.ymax - params.ymin)],
        }
        if repeats > 1:
            out.update(stats([n * steps / t for t in times]))
            out["values"] = [n * steps / t for t in times]
        out["primed_steps"] = prime
        return out


def brief(d):
    """An extra scene's entry: the headline numbers without the long roofline notes."""
    o = {k: d[k] for k in ("workload", "particles", "grid", "steps", "warmup", "evolved_steps", "value", "unit", "ms_per_step",
                           "mean_neighbors", "max_neighbors", "pair_interactions_per_s", "tile_rounds", "pressure_paths") if k in d}
    o["roofline"] = {k: d["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms")}
    if d.get("roofline_valu"):
        o["roofline_valu"] = {k: d["roofline_valu"][k] for k in ("achieved", "peak", "unit", "frac")}
    return o


def workload_text(kind, n_total, rows, params):
    return "2D %s, %d particles (%d x %d lattice at spacing H/4, box %.1f x %.1f, sph.h constants), fp32" % (
        "dam-break" if kind == "dam" else "uniform box filled wall to wall", n_total, n_total // rows, rows,
        params.xmax - params.xmin, params.ymax - params.ymin)


# ---- N > 1, one rank per GPU over torch.distributed (sph_amd/slab.py) -------------------------------------------------
def run_slabs(name, scaling, steps, warmup, rebalance_every, local_rank, rank, world, prime=0):
    """One slab-decomposed scene on the ranks of the default process group; returns the result dict on rank 0 (None
    elsewhere).  Timed region: barrier + device sync on both sides, MAX over ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from sph_amd import lib as _lib
    from sph_amd import scenes
    from sph_amd.slab import HALO_COLS, PACK_REACH, make_hip_slab

    kind, cols, rows = WORKLOADS[name]
    # weak scaling: the fluid column gets `world` times wider, per-GPU particle count fixed;
    # strong scaling: the named scene as it is, split into `world` slabs
    total_cols = cols * world if scaling == "weak" else cols
    sharded = scenes.dam_break_sharded if kind == "dam" else scenes.uniform_box_sharded
    params, n_total, hist_fn, make_local = sharded(total_cols, rows)
    drv = _lib.derive(params)
    slab = make_hip_slab(params, make_local, local_rank, hist=hist_fn(drv))
    inner = slab.solver
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def allmax(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ints(x):
        t = torch.zeros(world, dtype=torch.int64, device=dev)
        t[rank] = int(x)
        dist.all_reduce(t)
        return [int(v) for v in t.cpu()]

    def owned_now():
        # particles in this rank's own columns (ghost copies lie outside them); synchronises, and raises if a halo
        # message or the particle capacity overflowed during the asynchronous steps; leaves the state as it is.
        # Collective: a rank that raised alone would leave the others waiting in the next all-reduce.
        failure, v = None, 0
        try:
            v = int(slab.backend.column_histogram()[slab.c0:slab.c1].sum())
        except Exception as e:  # re-raised below, on every rank
            failure = e
        bad = allmax(1.0 if failure is not None else 0.0)
        if failure is not None:
            raise failure
        if bad:
            raise RuntimeError("another rank reported a failure (halo message or particle capacity overflow)")
        return v

    primer = None
    if prime:
        # set-up, as at N = 1 (run_single): a separate small scene on this rank's GPU, stepped right before the timed
        # region so that the GPU enters it at its clocks; the slab's own state sees exactly `warmup` steps
        from sph_amd.solver import Solver

        p1, a1 = scenes.dam_break(1024, 1024)
        primer = Solver(p1, n_max=len(a1), device=local_rank)
        primer.upload(a1)
        del a1
    slab.run(warmup, rebalance_every)
    torch.cuda.synchronize()
    counts_start = gather_ints(owned_now())
    if primer is not None:
        primer.stepSimulation(nsteps=prime)
        primer.synchronize()
        primer.close()
    # HIP events around the two force launches of a split step (boundary columns on the library's stream, interior
    # columns on its side stream), one step in eight
    inner.set_profiling_phases(["force", "force_interior"])
    inner.set_profiling_period(FORCE_SAMPLE_PERIOD)
    inner.reset_phase_times()
    reb0 = slab.rebalances
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slab.run(steps, rebalance_every)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = allmax(time.perf_counter() - t0)
    ph = inner.phase_times()
    inner.set_profiling(False)
    n_owned = owned_now()
    counts_end = gather_ints(n_owned)
    bytes_per_exchange = slab.bytes_per_exchange()
    # breakdown of 5 more steps with every phase bracketed (not part of `value`)
    inner.set_profiling_period(1)
    inner.set_profiling(True)
    inner.reset_phase_times()
    slab.step(5)
    torch.cuda.synchronize()
    breakdown = {k: v[0] / 5 for k, v in inner.phase_times().items() if v[1]}
    inner.set_profiling(False)
    # the interior launch is the dominant kernel: particles it updates = owned minus the PACK_REACH columns per neighbour
    slab.backend.drop_ghosts()
    h = slab.backend.column_histogram()
    il = slab.c0 + (PACK_REACH if rank > 0 else 0)
    ih = slab.c1 - (PACK_REACH if rank < world - 1 else 0)
    n_interior = int(h[il:ih].sum())
    inner.set_owned_columns(0, drv.nx)  # count pairs over this rank's own particles only (edge pairs are missed)
    inner.computeDensityPressure()
    pairs = float(inner.pair_count())
    t = torch.tensor([pairs], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    pairs = float(t.item())
    fi_ms, fi_k = ph["force_interior"]
    fe_ms, fe_k = ph["force"]
    interior_s = fi_ms / max(fi_k, 1) * 1e-3
    if rank != 0:
        del slab
        return None
    if fi_k:
        hbm, _ = force_rooflines(n_interior, interior_s, 0, 0, kernel="k_force<fused integrate>, interior columns of rank 0's slab")
        hbm["boundary_launch_ms"] = fe_ms / max(fe_k, 1)
        hbm["note"] = ("split slab step: this launch runs on the side stream beside the boundary columns' launch (boundary_launch_ms); "
                       "sampled one step in %d inside the timed region. " % FORCE_SAMPLE_PERIOD) + hbm["note"]
    else:  # (one rank, or steps too few to sample a split step: the whole pass of the un-split steps)
        hbm, _ = force_rooflines(n_owned, fe_ms / max(fe_k, 1) * 1e-3, 0, 0)
    out = {
        "workload": name, "scaling": scaling, "particles": n_total, "grid": [drv.nx, drv.ny], "steps": steps, "warmup": warmup,
        "value": n_total * steps / elapsed, "unit": "particle-updates/s", "ms_per_step": elapsed / steps * 1e3,
        "pair_interactions_per_s": 2.0 * pairs * steps / elapsed, "mean_neighbors": pairs / max(n_total, 1),
        "roofline": hbm, "phases_ms_per_step_rank0": breakdown,
        "slabs": {
            "bounds": list(slab.bounds), "particles_per_rank_start": counts_start, "particles_per_rank_end": counts_end,
            "rebalance_every": rebalance_every, "rebalances": slab.rebalances - reb0,
            "transport": "c10d-" + dist.get_backend() + (" (messages staged through host memory)" if dist.get_backend() != "nccl" else ""),
            "halo_columns": HALO_COLS, "message_records_steady": [c for c in slab.caps[1:-1]],
            "bytes_sent_per_exchange_rank0": bytes_per_exchange,
        },
        "workload_text": workload_text(kind, n_total, rows, params), "primed_steps": prime,
    }
    del slab
    return out


def main_ranks(args, world, rank, local_rank):
    """N > 1 under torch.distributed (one rank per GPU)."""
    import torch
    import torch.distributed as dist

    share = os.environ.get("SPH_BENCH_SHARE_GPU") == "1"  # rehearsal on a one-GPU box: every rank on device 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if share:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (SPH_BENCH_SHARE_GPU=1 rehearses the protocol with every "
                         "slab on device 0)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("SPH_BENCH_BACKEND", "gloo" if share else "nccl")  # RCCL refuses two ranks on one GPU
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    name = args.workload or "dam16m"
    scaling = args.scaling or ("strong" if args.workload is None else "weak")
    reb = int(os.environ.get("SPH_BENCH_REBALANCE", str(args.rebalance_every)))
    head = run_slabs(name, scaling, args.steps, args.warmup, reb, local_rank, rank, world, prime=args.prime)
    extras = {}
    if not args.no_extra_scenes and args.workload is None:
        # configs[4]: 8M particles per GPU, weak scaling (fewer steps: a bounded addition to the run)
        st = max(args.steps // 2, 5)
        try:
            r = run_slabs("dam8m", "weak", st, max(args.warmup // 2, 2), reb, local_rank, rank, world, prime=args.prime)
            if rank == 0:
                r["roofline"] = {k: r["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms")}
                extras["dam8m_per_gpu_weak"] = r
        except Exception as e:  # an extra must not take the headline down (every rank fails alike: sizes are global)
            if rank == 0:
                extras["dam8m_per_gpu_weak"] = {"error": str(e)}
    if rank == 0:
        out = {
            "metric": "particle-updates/sec (full SPH step)", "value": head["value"], "unit": "particle-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": head["workload_text"], "name": name, "particles": head["particles"], "grid": head["grid"],
                "parallelism": "x-slabs x%d, 2-column ghost halo + migration, one rank per GPU (torch.distributed)" % world,
                "driver": "c10d", "slabs": head["slabs"],
                "semantics": "snapshot (Jacobi) neighbour velocities, as SPH/MPI_sph.cpp",
            },
            "roofline": head["roofline"], "roofline_valu": None,
            "phases_ms_per_step": head["phases_ms_per_step_rank0"],
            "pair_interactions_per_s": head["pair_interactions_per_s"], "mean_neighbors": head["mean_neighbors"],
            "ups": args.steps / (head["ms_per_step"] * 1e-3 * args.steps), "primed_steps": head["primed_steps"],
        }
        if not args.no_extra_scenes and scaling == "strong":
            # the same scene on ONE GPU (rank 0's), for the strong-scaling ratio and the fp32-vector roofline of the kernel
            try:
                one = run_single(name, max(args.steps // 4, 5), 3, local_rank)
                extras[name + "_1gpu"] = brief(one)
                out["roofline_valu"] = one["roofline_valu"]
                if out["roofline_valu"]:
                    out["roofline_valu"]["note"] = "measured on the single-GPU run of the same scene (extra_scenes.%s_1gpu)" % name
            except Exception as e:
                extras[name + "_1gpu"] = {"error": str(e)}
        if extras:
            out["extra_scenes"] = extras
        if not args.no_cpu_baseline:
            base, extra = cpu_baseline()
            out["cpu_baseline"] = base
            out.update(extra)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ---- N > 1 in ONE process: the C++ driver libsph_mg.so --------------------------------------------------------------
def main_mg(args):
    import numpy as np
    import torch

    from sph_amd import scenes
    from sph_amd.mg import MultiGpuSolver

    world = args.gpus
    share = os.environ.get("SPH_BENCH_SHARE_GPU") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if not share and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d --driver mg: only %d GPU(s) visible (SPH_BENCH_SHARE_GPU=1 rehearses the protocol "
                         "with every slab on device 0)" % (world, torch.cuda.device_count()))
    devices = [0] * world if share else list(range(world))
    reb = int(os.environ.get("SPH_BENCH_REBALANCE", str(args.rebalance_every)))

    def run(name, scaling, steps, warmup):
        kind, cols, rows = WORKLOADS[name]
        total_cols = cols * world if scaling == "weak" else cols
        if kind == "dam":
            params, n_total, _, make_local = scenes.dam_break_sharded(total_cols, rows)
        else:
            params, n_total, _, make_local = scenes.uniform_box_sharded(total_cols, rows)
        from sph_amd import lib as _lib

        drv = _lib.derive(params)
        a, ids = make_local(0, drv.nx, drv)
        full = np.zeros(n_total, a.dtype)
        full[ids] = a
        del a, ids
        with MultiGpuSolver(params, full, devices=devices) as m:
            del full
            counts_start = m.counts()
            m.step(warmup, rebalance_every=reb)
            m.synchronize()
            r0 = m.rebalances
            t0 = time.perf_counter()
            m.step(steps, rebalance_every=reb)
            m.synchronize()
            elapsed = time.perf_counter() - t0
            hs = m.halo_stats()
            return {
                "workload": name, "scaling": scaling, "particles": n_total, "grid": [drv.nx, drv.ny], "steps": steps, "warmup": warmup,
                "value": n_total * steps / elapsed, "unit": "particle-updates/s", "ms_per_step": elapsed / steps * 1e3,
                "slabs": {"bounds": m.bounds(), "particles_per_rank_start": counts_start, "particles_per_rank_end": m.counts(),
                          "rebalance_every": reb, "rebalances": m.rebalances - r0, "transport": m.transport,
                          "bytes_sent_per_exchange_per_slab": hs["bytes_per_exchange_per_slab"],
                          "message_records_steady_max": hs["max_message_records"]},
                "workload_text": workload_text(kind, n_total, rows, params),
            }

    name = args.workload or "dam16m"
    scaling = args.scaling or ("strong" if args.workload is None else "weak")
    head = run(name, scaling, args.steps, args.warmup)
    out = {
        "metric": "particle-updates/sec (full SPH step)", "value": head["value"], "unit": "particle-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head["workload_text"], "name": name, "particles": head["particles"], "grid": head["grid"],
                   "parallelism": "x-slabs x%d, 2-column ghost halo + migration, ONE process (libsph_mg.so: ncclCommInitAll, one "
                                  "enqueue thread per slab)" % world,
                   "driver": "mg", "slabs": head["slabs"], "semantics": "snapshot (Jacobi) neighbour velocities, as SPH/MPI_sph.cpp"},
    }
    extras = {}
    if not args.no_extra_scenes and args.workload is None:
        try:
            extras["dam8m_per_gpu_weak"] = run("dam8m", "weak", max(args.steps // 2, 5), max(args.warmup // 2, 2))
        except Exception as e:
            extras["dam8m_per_gpu_weak"] = {"error": str(e)}
    if not args.no_extra_scenes and scaling == "strong":
        try:
            one = run_single(name, max(args.steps // 4, 5), 3, devices[0])
            extras[name + "_1gpu"] = brief(one)
            # (the C++ driver exposes no per-kernel events: the kernel's rooflines are those of the single-GPU run)
            out["roofline"] = one["roofline"]
            out["roofline"]["note"] = "measured on the single-GPU run of the same scene (extra_scenes.%s_1gpu). " % name + out["roofline"]["note"]
            out["roofline_valu"] = one["roofline_valu"]
        except Exception as e:
            extras[name + "_1gpu"] = {"error": str(e)}
    if extras:
        out["extra_scenes"] = extras
    if not args.no_cpu_baseline:
        base, extra = cpu_baseline()
        out["cpu_baseline"] = base
        out.update(extra)
    print(json.dumps(out), flush=True)


# ---- N = 1 ----------------------------------------------------------------------------------------------------------
def main_single(args):
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    device = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    name = args.workload or "dam1m"
    kind, cols, rows = WORKLOADS[name]
    head = run_single(name, args.steps, args.warmup, device, repeats=args.repeats, prime=args.prime)
    out = {
        "metric": "particle-updates/sec (full SPH step)", "value": head["value"], "unit": "particle-updates/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "2D %s, %d particles (%d x %d lattice at spacing H/4, box %.1f x %.1f, sph.h constants), fp32" % (
                "dam-break" if kind == "dam" else "uniform box filled wall to wall", head["particles"], head["particles"] // rows, rows,
                head["box"][0], head["box"][1]),
            "name": name, "particles": head["particles"], "grid": head["grid"], "parallelism": "single GPU",
            "semantics": "snapshot (Jacobi) neighbour velocities, as SPH/MPI_sph.cpp",
        },
        "roofline": head["roofline"], "roofline_valu": head["roofline_valu"],
        "phases_ms_per_step": head["phases_ms_per_step"],
        "pair_interactions_per_s": head["pair_interactions_per_s"], "mean_neighbors": head["mean_neighbors"],
        "ups": args.steps / (head["ms_per_step"] * 1e-3 * args.steps), "tile_rounds": head["tile_rounds"],
        "pressure_paths": head["pressure_paths"],
    }
    for k in ("repeats", "value_median", "value_min", "value_max", "spread", "values", "primed_steps"):
        if k in head:
            out[k] = head[k]
    if not args.no_extra_scenes:
        # the other single-GPU scenes of BASELINE.json (configs[2], the scene of configs[3]) and the headline scene
        # after 1000 untimed steps (collapsed column, cells up to twice their initial population), same run
        extras = {}
        for key, wl, st, evo in (("dam1m_evolved", name, args.steps, 1000), ("box4m", "box4m", max(args.steps // 2, 5), 0),
                                 ("dam16m", "dam16m", max(args.steps // 4, 5), 0)):
            try:
                extras[key] = brief(run_single(wl, st, max(args.warmup, 3), device, evolve=evo))
            except Exception as e:  # an extra must not take the headline down
                extras[key] = {"error": str(e)}
        out["extra_scenes"] = extras
    if not args.no_cpu_baseline:
        base, extra = cpu_baseline()
        out["cpu_baseline"] = base
        out.update(extra)
    print(json.dumps(out), flush=True)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (torch.distributed.run)
    BEFORE this process touches the GPU, relay their output, exit with their code."""
    import torch  # (importing torch and counting devices do not initialise the GPU)

    share = os.environ.get("SPH_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if have < args.gpus and not share:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; SPH_BENCH_SHARE_GPU=1 rehearses the protocol with every slab "
                         "on device 0" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16)))))  # (the launcher would set 1: the CPU baseline needs cores)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: dam1m at N = 1 (configs[1]); dam16m split over the GPUs at N > 1 (configs[3])")
    ap.add_argument("--rebalance-every", type=int, default=100,
                    help="N > 1: steps between two slab rebalances (0: never); also SPH_BENCH_REBALANCE")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-scenes", action="store_true",
                    help="skip the extra scenes reported beside the headline (N = 1: box4m, dam16m, evolved dam1m; N > 1: dam8m "
                    "per GPU weak scaling, the headline scene on one GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong = the named workload split over the GPUs (default without --workload: dam16m, "
                    "BASELINE configs[3]); weak = the named workload per GPU (default with --workload)")
    ap.add_argument("--driver", choices=["c10d", "mg"], default="c10d",
                    help="N > 1: c10d = one process per GPU over torch.distributed (sph_amd/slab.py); mg = one process, "
                    "libsph_mg.so (C++ driver, ncclCommInitAll)")
    ap.add_argument("--prime", type=int, default=150, help="set-up steps that bring the GPU to its clocks before the measured scene "
                    "does its --warmup untimed and --steps timed steps: at N = 1 on the scene itself, which is then uploaded "
                    "again; at N > 1 (c10d driver) on a separate 1M scene on each rank's GPU, right before the timed region")
    ap.add_argument("--repeats", type=int, default=5, help="N = 1: timed intervals of --steps steps (value = the first; "
                    "median / min / max over all of them are reported beside it)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus == 1 and (env_world is None or env_world == "1") and os.environ.get("SPH_BENCH_FORCE_SLAB") != "1":
        return main_single(args)
    if args.driver == "mg":
        if env_world not in (None, "1"):
            raise SystemExit("--driver mg runs all slabs from ONE process: start it without torch.distributed.run")
        return main_mg(args)
    if env_world is None:
        return self_launch(args, sys.argv[1:])
    world = int(env_world)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    return main_ranks(args, world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")))


if __name__ == "__main__":
    main()
