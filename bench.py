#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the full SPH step (cell list + density/pressure +
force + integrate/boundary) with the state resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dam1m|box4m|dam16m|...]

N = 1: BASELINE.json configs[1], the 2-D dam break with 1M particles (1024 x 1024 lattice at
spacing H/4 against the left wall of a 4W x 2W box, all sph.h constants).  N > 1 (launched by
torch.distributed.run, one rank per GPU): x-slab decomposition with ghost-column halo exchange and
migration over RCCL (sph_amd/slab.py); per-GPU work is fixed (weak scaling): the column gets N
times wider.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (fused force + integrate)
by its ALGORITHMIC bytes (40 B/particle: R pos 8, vel 8, rho 4, P 4; W pos 8, vel 8; neighbour
gathers are served by LDS/L2 and not counted) over its average launch time,
measured with HIP events on the library's stream inside the timed region.  `cpu_baseline` is the
oracle's portable restatement of the reference's OpenMP path (SPH/openMP_sph.cpp:80-205,
all-pairs) timed on this host's cores on a bounded miniature of the same scene.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sph_amd import scenes  # noqa: E402
from sph_amd.solver import Solver  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X peak FP32 vector (MI355X_MICROARCH.md)
# HIP events around k_force in one step of this many inside the timed region: an event pair leaves ~11 us of gaps on the
# stream (rocprofv3 kernel trace), i.e. bracketing every step would slow the measured loop itself by 4-5 %
FORCE_SAMPLE_PERIOD = 8
FORCE_BYTES_PER_PARTICLE = 40  # fused force + integrate, SURVEY.md 8(d)
# algorithmic flops, SURVEY.md 8(d): per stencil candidate 7 (2 sub, 2 mul, add, cmp, select), per in-range density
# pair +6, per in-range force pair +45 (div and sqrt counted as 1 each), per particle epilogue ~25
FLOP_CANDIDATE, FLOP_DENSITY_PAIR, FLOP_FORCE_PAIR, FLOP_EPILOGUE = 7, 6, 45, 25

WORKLOADS = {
    # name: (kind, lattice columns per GPU, rows)
    "dam1m": ("dam", 1024, 1024),
    "dam4m": ("dam", 2048, 2048),
    "dam16m": ("dam", 4096, 4096),
    "dam8m": ("dam", 2896, 2896),
    "box4m": ("box", 2048, 2048),
    "dam64k": ("dam", 256, 256),
}


def cpu_baseline(n_gpus):
    """Times the oracle's OpenMP all-pairs port (and the cell-list oracle) on this host."""
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
    return base, extra


def stencil_candidates(solver, drv):
    """Sum over particles of the population of their 3x3 cell stencil (the candidates a cell-list pass tests), from
    the cell table of the current grid.  Host numpy, off the timed path."""
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


def force_rooflines(n_local, force_avg_s, pairs_local, cand_local, traffic=None, valu=None):
    """The two roofline blocks of the dominant kernel (fused force + integrate): HBM by algorithmic bytes, and FP32
    vector by SURVEY.md 8(d)'s algorithmic flop counts (the bound that actually applies at ~50 neighbours)."""
    achieved = FORCE_BYTES_PER_PARTICLE * n_local / force_avg_s / 1e9 if force_avg_s > 0 else 0.0
    hbm = {
        "kernel": "k_force<fused integrate>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": force_avg_s * 1e3,
        "algorithmic_bytes_per_launch": FORCE_BYTES_PER_PARTICLE * n_local, "valu": valu,
        "note": "pair kernels are VALU-bound at ~50 neighbours (no MFMA: irregular gathers); see roofline_valu and DESIGN.md",
    }
    flops = FLOP_CANDIDATE * cand_local + FLOP_FORCE_PAIR * pairs_local + FLOP_EPILOGUE * n_local
    tf = flops / force_avg_s / 1e12 if force_avg_s > 0 else 0.0
    fp32 = {
        "kernel": "k_force<fused integrate>", "bound": "valu_fp32", "achieved": tf, "peak": FP32_VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TFLOPS, "flops_per_launch": flops,
        "model": "SURVEY.md 8(d): 7 per stencil candidate + 45 per in-range pair + 25 per particle; candidates %d, in-range "
                 "pairs %d" % (cand_local, pairs_local),
    }
    return hbm, fp32


def run_single(name, steps, warmup, device, evolve=0):
    """One single-GPU scene: `evolve` untimed steps, `warmup` untimed steps, `steps` timed steps with HIP events around
    the force kernel, then neighbour statistics.  Returns a dict (and the solver's scene description)."""
    kind, cols, rows = WORKLOADS[name]
    params, a = scenes.dam_break(cols, rows) if kind == "dam" else scenes.uniform_box(cols)
    n = len(a)
    with Solver(params, n_max=n, device=device) as solver:
        solver.upload(a)
        del a
        if evolve:
            solver.stepSimulation(nsteps=evolve)
        solver.stepSimulation(nsteps=warmup)
        torch.cuda.synchronize()
        solver.set_profiling_phases(["force"])
        solver.set_profiling_period(FORCE_SAMPLE_PERIOD)
        solver.reset_phase_times()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.stepSimulation(nsteps=steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        f_ms, f_k = solver.phase_times()["force"]
        solver.set_profiling(False)
        solver.computeDensityPressure()
        pairs = solver.pair_count()
        counts = solver.neighbor_counts()
        cand = stencil_candidates(solver, solver.drv)
        force_avg_s = (f_ms / max(f_k, 1)) * 1e-3  # (f_k: the sampled launches)
        hbm, fp32 = force_rooflines(n, force_avg_s, pairs, cand)
        return {
            "workload": name, "particles": n, "grid": [solver.drv.nx, solver.drv.ny], "steps": steps, "warmup": warmup,
            "evolved_steps": evolve, "value": n * steps / elapsed, "unit": "particle-updates/s",
            "ms_per_step": elapsed / steps * 1e3, "mean_neighbors": pairs / max(n, 1),
            "max_neighbors": int(counts.max()) - 1 if n else 0,
            "pair_interactions_per_s": 2.0 * pairs * steps / elapsed,
            "roofline": {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms")},
            "roofline_valu": {k: fp32[k] for k in ("achieved", "peak", "unit", "frac")},
        }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--rebalance-every", type=int, default=100,
                    help="N > 1: steps between two slab rebalances (0: never); also SPH_BENCH_REBALANCE")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-scenes", action="store_true",
                    help="N = 1: skip the extra single-GPU scenes (box4m, dam16m, evolved dam1m) reported beside the headline")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = the workload per GPU (default, the driver's contract); strong = the named workload "
                    "split over the GPUs (BASELINE configs[3]: --workload dam16m --scaling strong)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("SPH_BENCH_SHARE_GPU") == "1":  # debugging on a one-GPU box: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)

    name = args.workload or "dam1m"
    kind, cols, rows = WORKLOADS[name]

    force_slab = os.environ.get("SPH_BENCH_FORCE_SLAB") == "1"  # exercise the slab path on one GPU
    if world == 1 and not force_slab:
        params, a = scenes.dam_break(cols, rows) if kind == "dam" else scenes.uniform_box(cols)
        n_total = len(a)
        solver = Solver(params, n_max=n_total, device=local_rank)
        solver.upload(a)
        step = lambda k: solver.stepSimulation(nsteps=k)  # noqa: E731
        sync = lambda: torch.cuda.synchronize()  # noqa: E731
        barrier = lambda: None  # noqa: E731
        parallelism = "single GPU"
        drv = solver.drv
    else:
        import torch.distributed as dist

        from sph_amd import lib as _lib
        from sph_amd.slab import make_hip_slab

        if world == 1:  # SPH_BENCH_FORCE_SLAB without a launcher: a one-rank group
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        backend = os.environ.get("SPH_BENCH_BACKEND", "nccl")  # "gloo": debugging on a one-GPU box (host-staged messages)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # weak scaling: the fluid column gets `world` times wider, per-GPU particle count fixed;
        # strong scaling: the named scene as it is, split into `world` slabs
        total_cols = cols * world if args.scaling == "weak" else cols
        sharded = scenes.dam_break_sharded if kind == "dam" else scenes.uniform_box_sharded
        params, n_total, hist_fn, make_local = sharded(total_cols, rows)
        drv = _lib.derive(params)
        solver = make_hip_slab(params, make_local, local_rank, hist=hist_fn(drv))
        # the dam collapses to the right: slabs are re-balanced (and the halo messages re-sized) every K steps
        rebalance_every = int(os.environ.get("SPH_BENCH_REBALANCE", str(args.rebalance_every)))
        step = lambda k: solver.run(k, rebalance_every)  # noqa: E731
        sync = lambda: torch.cuda.synchronize()  # noqa: E731
        barrier = lambda: dist.barrier()  # noqa: E731
        parallelism = "x-slabs x%d, 2-column ghost halo + migration over RCCL" % world

    step(args.warmup)
    sync()
    slabbed = world > 1 or force_slab
    inner = solver.solver if slabbed else solver
    # HIP events around the dominant kernel only (events between the small back-to-back kernels
    # of the sort cost ~10 % of a step); the full per-phase breakdown is taken after the timed region
    # (not on slabs: a profiled slab step runs its two force passes back to back instead of side by side, and the
    # events would time the boundary pass only; the slab line takes the kernel time from the breakdown steps below)
    if not slabbed:
        inner.set_profiling_phases(["force"])
        inner.set_profiling_period(FORCE_SAMPLE_PERIOD)
    inner.reset_phase_times()
    barrier()
    sync()
    t0 = time.perf_counter()
    step(args.steps)
    sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    phases = inner.phase_times()
    inner.set_profiling(False)
    # per-phase breakdown: a few extra steps with every phase bracketed (not part of `value`)
    inner.set_profiling_period(1)
    inner.set_profiling(True)
    inner.reset_phase_times()
    step(5)
    sync()
    breakdown = inner.phase_times()
    inner.set_profiling(False)
    if slabbed:
        phases = breakdown
    if slabbed:
        solver.check()  # raises if a halo buffer / the particle capacity overflowed during the asynchronous steps
    n_local = inner.n
    if world > 1:
        # particles this rank integrates per step (ghosts excluded): what the force kernel's
        # algorithmic bytes are counted on
        n_local = n_total // world

    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # neighbour statistics, off the timed path
    if slabbed:
        inner.drop_ghosts()
        inner.set_owned_columns(0, drv.nx)  # count over this rank's own particles only (edge pairs are missed)
    inner.computeDensityPressure()
    pairs_local = inner.pair_count()
    cand_local = stencil_candidates(inner, drv) if world == 1 else 0
    if world > 1:
        t = torch.tensor([pairs_local], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        pairs = float(t.item())
    else:
        pairs = float(pairs_local)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        f_ms, f_k = phases["force"]
        # per step, not per launch: a slab step with neighbours runs the pass as two launches (boundary columns, interior)
        # single GPU: the sampled launches of the timed region; slabs: per step over the 5 breakdown steps (two launches each)
        force_avg_s = (f_ms / max(min(f_k, 5) if slabbed else f_k, 1)) * 1e-3
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(name, {}).get("k_force_bytes_per_launch")
            except Exception:
                traffic = None
        valu = None  # what actually bounds the kernel: VALU issue (PMC of the committed profile, informational)
        try:
            src = json.load(open(tf)).get(name, {}).get("source")
            pm = json.load(open(os.path.join(ROOT, src)))
            kf = [v for k, v in pm.items() if k.startswith("k_force<true")][0]
            valu = {"cycles_per_wave_instruction": kf.get("valu_cycles_per_wave_instr"), "lane_utilisation": kf.get("valu_lane_utilisation"),
                    "wave_instructions_per_launch": kf.get("SQ_INSTS_VALU"), "source": src}
        except Exception:
            valu = None
        roof_hbm, roof_fp32 = force_rooflines(n_local, force_avg_s, pairs_local, cand_local, traffic, valu)
        out = {
            "metric": "particle-updates/sec (full SPH step)",
            "value": value,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "2D %s, %%d particles (%%d x %%d lattice at spacing H/4, box %%.1f x %%.1f, sph.h constants), fp32"
                % ("dam-break" if kind == "dam" else "uniform box filled wall to wall")
                % (n_total, n_total // rows, rows, params.xmax - params.xmin, params.ymax - params.ymin),
                "name": name,
                "particles": n_total,
                "grid": [drv.nx, drv.ny],
                "parallelism": parallelism,
                "rebalance_every": (rebalance_every if (world > 1 or force_slab) else None),
                "rebalances": (solver.rebalances if (world > 1 or force_slab) else None),
                "semantics": "snapshot (Jacobi) neighbour velocities, as SPH/MPI_sph.cpp",
            },
            "roofline": roof_hbm,
            "roofline_valu": roof_fp32 if world == 1 else None,
            "phases_ms_per_step": {k: v[0] / 5 for k, v in breakdown.items()},
            "pair_interactions_per_s": 2.0 * pairs * args.steps / elapsed,  # density + force passes
            "mean_neighbors": pairs / max(n_total, 1),
            "ups": args.steps / elapsed,
        }
        if world == 1 and not force_slab and not args.no_extra_scenes:
            # the other single-GPU scenes of BASELINE.json (configs[2], the scene of configs[3]) and the headline scene
            # after 1000 untimed steps (collapsed column, cells up to twice their initial population), same run
            del solver
            extras = {}
            for key, wl, st, evo in (("dam1m_evolved", name, args.steps, 1000), ("box4m", "box4m", max(args.steps // 2, 5), 0),
                                     ("dam16m", "dam16m", max(args.steps // 4, 5), 0)):
                try:
                    extras[key] = run_single(wl, st, max(args.warmup, 3), local_rank, evolve=evo)
                except Exception as e:  # an extra must not take the headline down
                    extras[key] = {"error": str(e)}
            out["extra_scenes"] = extras
        if not args.no_cpu_baseline and world == 1:
            base, extra = cpu_baseline(world)
            out["cpu_baseline"] = base
            out.update(extra)
        print(json.dumps(out), flush=True)

    if slabbed:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
