#!/bin/bash
# tools/ab_scenes.sh VAR=val [VAR2=val2 ...] : the default bench line (headline + evolved + box4m + dam16m, no CPU legs) with
# and without each environment setting
for v in base "$@"; do
  if [ "$v" = "base" ]; then e=""; else e="$v"; fi
  out=$(env $e timeout -k 10 200 python bench.py --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | tail -1) || { echo "$v => failed"; exit 1; }
  echo "$v => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); x=d["extra_scenes"]; print("dam1m %.3e (median %.3e) evolved %.3e box4m %.3e dam16m %.3e"%(d["value"],d["value_median"],x["dam1m_evolved"]["value"],x["box4m"]["value"],x["dam16m"]["value"]))')"
done
