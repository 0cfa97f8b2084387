#!/bin/bash
# tools/rehearse.sh: the N > 1 bench protocol on a ONE-GPU box (every slab on device 0, messages over gloo): checks the
# launcher paths, slab bounds, message sizes and capacities at 4 and 8 ranks.  Not a measurement.
set -o pipefail
export SPH_BENCH_SHARE_GPU=1
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/reh4.json 2> gpurun_out/reh4.err || { echo "N=4 failed"; tail -c 1500 gpurun_out/reh4.err; exit 1; }
echo done4
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/reh8.json 2> gpurun_out/reh8.err || { echo "N=8 failed"; tail -c 1500 gpurun_out/reh8.err; exit 1; }
echo done8
python - <<EOF
import json
for f in ("gpurun_out/reh4.json", "gpurun_out/reh8.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    s = d["config"]["slabs"]
    print(f, d["n_gpus"], "%.3e" % d["value"], "%.3f ms" % d["ms_per_step"], s["particles_per_rank_end"], s["message_records_steady"],
          {k: (v.get("value"), v.get("error")) for k, v in d.get("extra_scenes", {}).items()})
EOF
