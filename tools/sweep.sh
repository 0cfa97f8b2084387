#!/bin/bash
# tools/sweep.sh "<env settings>" ... : bench one line per setting (force/density ms, total)
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-scenes 2>/dev/null | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d["phases_ms_per_step"]; print("value %.3e ms %.3f force %.3f density %.3f"%(d["value"],d["ms_per_step"],p["force"],p["density"]))')"
done
