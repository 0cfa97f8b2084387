#!/bin/bash
# quick SQ counter pass for the pair kernels:  tools/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-scenes $* > $OUT/pmc_sq1.log 2>&1 || { tail -5 $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-scenes $* > $OUT/pmc_sq2.log 2>&1 || { tail -5 $OUT/pmc_sq2.log; exit 1; }
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_sq*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if k.startswith("k_force") or k.startswith("k_density"):
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in d.items():
    m={c:sum(v)/len(v) for c,v in cs.items()}
    print(k)
    for c in sorted(m): print("   %-28s %14.0f"%(c,m[c]))
    if "SQ_THREAD_CYCLES_VALU" in m: print("   lane_util %.3f  valu_busy %.3f"%(m["SQ_THREAD_CYCLES_VALU"]/(m["SQ_ACTIVE_INST_VALU"]*64), m["SQ_ACTIVE_INST_VALU"]*4/(m["GRBM_GUI_ACTIVE"]/8*1024)))
PY
