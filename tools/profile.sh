#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun):
#   tools/profile.sh <tag> [bench args...]
# kernel-trace/stats in one run, then each PMC group in a run of its own (never combined with
# tracing), each under its own hard time limit.  Outputs under gpurun_out/<tag>/.
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extra-scenes $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {  # name, counters
  timeout -s KILL 180 rocprofv3 --pmc $2 --output-format csv -d "$OUT/pmc_$1" -o p -- python3 bench.py $ARGS > "$OUT/pmc_$1.log" 2>&1 || { echo "pmc pass $1 failed"; tail -5 "$OUT/pmc_$1.log"; exit 1; }
}
pass fetch "FETCH_SIZE" && \
pass write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" && \
pass sq1 "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" && \
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
echo "profile $TAG done"; ls "$OUT"
