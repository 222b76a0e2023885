#!/bin/bash
# Vector instructions per launch of the dominant kernel against the lane-refill threshold (tuning build; PT_REFILL is
# read by the one-launch kernel, PT_REFILL_AT by the traversal kernel):
#   tools/refill_sweep.sh LIB [bench args]
LIB=$1; shift
for v in 8 12 16 24 32 48; do
  PT_REFILL=$v PT_REFILL_AT=$v PT_AMD_LIB=$PWD/$LIB bash tools/pmc_pass.sh refill_$v "SQ_INSTS_VALU SQ_INSTS_SALU" "$@" > /dev/null 2>&1
  echo "refill $v: $(grep -E 'k_bounces|k_extend_bvh<[0-9]+, false, false' gpurun_out/pmc/refill_$v.txt | head -1)"
done
