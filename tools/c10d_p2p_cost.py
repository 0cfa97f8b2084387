"""Host cost of one batch_isend_irecv (nccl) as sph_amd/slab.py issues it, on a one-rank group sending to itself.
    python tools/c10d_p2p_cost.py"""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
cap = 20560
a, b, c, d = (torch.zeros((cap + 1, 5), dtype=torch.int32, device=dev) for _ in range(4))
a += 7
try:
    ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
    for _ in range(5):
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        works = dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch_isend_irecv (2 sends + 2 recvs of %d KB to self) + wait: host %.1f us per call, wall %.1f us; received ok: %s" % (
        a.numel() * 4 // 1024, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, bool((b == 7).all())), flush=True)
except Exception as e:
    print("self send/recv not possible here:", repr(e)[:300], flush=True)
dist.destroy_process_group()
