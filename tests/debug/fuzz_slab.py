"""Test infrastructure (uses the oracle): random scenes through the slab path with the HIP backend, 2-4 ranks
sharing the GPU of the test box (messages staged through gloo), against the single-GPU run and the oracle:
    python tests/debug/fuzz_slab.py <first seed> <last seed + 1>"""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from sph_amd import scenes


def scene(seed):
    rng = np.random.default_rng(9000 + seed)
    if seed % 3 == 0:  # reference box, dense
        from sph_amd.params import SphParams
        params, a = SphParams.reference(), scenes.lattice(int(rng.choice([2000, 5000, 8192])))
    elif seed % 3 == 1:
        params, a = scenes.dam_break(int(rng.choice([48, 96, 128])), int(rng.choice([16, 40, 64])))
        a["pos"][:, 0] += np.float32(rng.uniform(0.0, 6.0))
    else:
        params, a = scenes.uniform_box(int(rng.choice([40, 72, 100])))
    a["pos"] += rng.uniform(-0.01, 0.01, size=a["pos"].shape).astype(np.float32)
    a["vel"][:, 0] = rng.normal(0.0, 3.0, size=len(a)).astype(np.float32) + np.float32(rng.choice([-4.0, 0.0, 4.0]))
    a["vel"][:, 1] = rng.normal(0.0, 1.0, size=len(a)).astype(np.float32)
    if seed % 4 == 0:
        a["valid"][::13] = 0
    world = int(rng.choice([2, 3, 4]))
    plan = [int(x) for x in rng.choice([1, 2, 5, 9, 17], size=3)]  # steps per call, a rebalance between calls
    return params, a, world, plan


def worker(rank, world, port, seed, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from sph_amd.slab import make_hip_slab

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params, a, _, plan = scene(seed)
        slab = make_hip_slab(params, a, device=0)
        for i, k in enumerate(plan):
            if i:
                slab.rebalance()
            slab.step(k)
        got = slab.gather(len(a))
        if rank == 0:
            np.save(out_path, got)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        params, a, world, plan = scene(seed)
        nx = __import__("sph_amd.lib", fromlist=["derive"]).derive(params).nx
        if nx < 3 * world + 2:
            world = 2
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = "/tmp/fuzz_slab_%d.npy" % seed
        want = orc.Oracle().step_cells(params, a.copy(), sum(plan))
        blown = bool(np.isnan(want["pos"]).any())  # the model itself produced NaN (LJ overflow of a very close pair)
        try:
            mp.spawn(worker, args=(world, port, seed, out), nprocs=world, join=True)
            got = np.load(out)
            ok = orc.same_bits_nan(got, want)
        except Exception as e:  # noqa: BLE001
            # a non-finite position has no column: the slab path reports it (sticky SPH_ERR_OUT_OF_SLAB) instead of
            # keeping the particle in cell 0 as the single-domain path and the oracle do
            ok = blown and "farther than the halo" in repr(e)
            print("seed", seed, "EXC (expected: NaN scene)" if ok else "EXC", repr(e)[-300:] if ok else repr(e)[-1500:], flush=True)
        print("seed", seed, "world", world, "n", len(a), "plan", plan, "ok" if ok else "MISMATCH", flush=True)
        if not ok:
            bad.append(seed)
    print("done; failures:", bad)
    sys.exit(1 if bad else 0)
