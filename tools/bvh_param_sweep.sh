#!/bin/bash
# Vector instructions per launch of the traversal kernels against the builder's leaf size and traversal cost (tuning
# build, host builder):  tools/bvh_param_sweep.sh LIB "LEAVES" "COSTS" [bench args]
LIB=$1; LEAVES=$2; COSTS=$3; shift 3
for leaf in $LEAVES; do for ct in $COSTS; do
  PT_BVH_LEAF=$leaf PT_BVH_CT=$ct PT_AMD_LIB=$PWD/$LIB bash tools/pmc_pass.sh bvh_${leaf}_$ct "SQ_INSTS_VALU" "$@" > /dev/null 2>&1
  echo "leaf $leaf ct $ct: value $(python3 -c "import json;print(round(json.loads(open('gpurun_out/pmc/bvh_${leaf}_$ct.json').read().strip().splitlines()[-1])['value']))")"
  grep -E 'k_bounces|k_extend_bvh' gpurun_out/pmc/bvh_${leaf}_$ct.txt
done; done
