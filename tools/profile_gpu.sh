#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes
# over a short bench.py run.  Results land in gpurun_out/prof/<tag>/.
# usage: tools/profile_gpu.sh TAG [bench args...]
set -u
TAG=${1:-profile}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
# PROG overrides the profiled command (default: a short bench.py run with the given arguments)
BENCH=${PROG:-"python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-alone $*"}
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL ${PASS_LIMIT:-150} rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
run kt --kernel-trace --stats || exit 1
# the same run with one batch at a time (one stream, every kernel alone on the GPU)
BENCH_KEEP=$BENCH; BENCH="$BENCH --batch-streams 1"
run kt_alone --kernel-trace --stats || exit 1
BENCH=$BENCH_KEEP
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE || exit 1
run pmc_sq3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY || exit 1
# how busy the CU's vector-memory address and data paths are (not fatal where the counters are not offered)
run pmc_tatd --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE || echo "(no TA/TD pass)"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
