C='SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU'
for v in n80 n112; do for sc in Box Box_Anisotropic; do
  export PT_AMD_LIB=$PWD/pathtracer_amd/variants/$v.so
  echo "== $v $sc"; bash tools/pmc_pass.sh lds_${v}_$sc "$C" --scene $sc | grep "extend_bvh<256, false, false" | cut -c1-500
done; done
