# The GPU suite under every switch that selects an alternative PATH (DESIGN.md "Switches": 14 rows; tuning parameters —
# task sizes, supernode widths, thread counts — are exercised by targeted tests, not by the matrix):
#   bash profiles/switch_matrix.sh > gpurun_out/switch_matrix.txt
for v in "SLPX_FUSE_LAUNCHES=0" "SLPX_FUSE_KKT=0" "SLPX_LDLT_MF=0" "SLPX_LDLT_IL=0" \
         "SLPX_TAPE_TEMPLATES=0" "SLPX_TAPE_JIT=0" "SLPX_CHAIN_TAPE=0" "SLPX_IPM_RESIDENT=0" "SLPX_IPM_LOOKAHEAD=0" \
         "SLPX_IPM_PIPELINE=0" "SLPX_IPM_RIDE=0" "SLPX_TWIN=0" "SLPX_MF_SOLVE=0" "SLPX_HESSIAN_FAMILIES=0"; do
  echo -n "$v: "
  env $v timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
done
