#!/bin/bash
# tools/check_change.sh <tag>: after a kernel change -- the whole GPU suite, the scene bench line, and a kernel trace of
# the 16M and the 1M scene (grid-build kernels listed).  Stops at the first failure.
TAG=${1:-chk}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -15 gpurun_out/${TAG}_tests.log; exit 1; }
tail -1 gpurun_out/${TAG}_tests.log
bash tools/ab_scenes.sh || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in dam16m dam1m; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$wl -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-scenes --workload $wl > gpurun_out/${TAG}_$wl.log 2>&1 || { echo "trace $wl failed"; exit 1; }
  echo "== $wl"; python3 - "$TAG" "$wl" <<PY
import csv, sys
tag, wl = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open("gpurun_out/%s_%s/t_kernel_stats.csv" % (tag, wl))):
    n = r["Name"].split("(")[0]
    if int(r["Calls"]) > 50: print("  %-28s %8.1f us" % (n[-28:], float(r["AverageNs"]) / 1e3))
PY
done
