for rep in 1 2; do
for v in "A=1" "SPHHIP_LONG_TILES=0" "SPHHIP_LONG_TILES=1" ; do
  out=$(env $v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-scenes --workload dam16m 2>/dev/null | tail -1)
  echo "$v => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); p=d["phases_ms_per_step"]; print("ms %.4f force %.4f density %.4f"%(d["ms_per_step"],p["force"],p["density"]))')"
done; done
