#!/bin/bash
# tools/ab_env.sh VAR=val [VAR2=val2 ...] : the bench line with and without each environment setting, alternating twice
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = "base" ]; then e=""; else e="$v"; fi
  out=$(env $e timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-scenes ${AB_ARGS:-} 2>/dev/null | tail -1)
  echo "$v => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d["phases_ms_per_step"]; print("value %.3e median %.3e ms %.4f force %.4f density %.4f grid %.4f (hash %.4f sort %.4f cells %.4f reorder %.4f)"%(d["value"],d.get("value_median",0),d["ms_per_step"],p["force"],p["density"],p["hash"]+p["sort"]+p["cells"]+p["reorder"],p["hash"],p["sort"],p["cells"],p["reorder"]))')"
done
done
