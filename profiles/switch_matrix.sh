# The GPU suite under every switch that selects an alternative path (DESIGN.md §4 "Switches"; r05: 22 rows — 20 path switches
# after the prune — the experiments measured as losses in r03/r04 are gone from the library, their record is in DESIGN.md):
#   bash profiles/switch_matrix.sh > gpurun_out/switch_matrix.txt
for v in "SLPX_FUSE_LAUNCHES=0" "SLPX_FUSE_KKT=0" "SLPX_FUSE_BACKSUB=0" "SLPX_FUSE_SOLVE=0" "SLPX_SUPERNODAL=0" \
         "SLPX_SINGLE_LAUNCH=0" "SLPX_TAPE_JIT=0" "SLPX_TAPE_SPECIALIZE=0" "SLPX_IPM_RESIDENT=0" "SLPX_LDLT_IL=0" \
         "SLPX_TAPE_CSE=0" "SLPX_LDLT_MF=0" "SLPX_RELAX_ZEROS=0" "SLPX_MFMA_MIN_ENTRIES=0" "SLPX_MF_THREADS=512" \
         "SLPX_CHAIN_TAPE=0" "SLPX_IPM_LOOKAHEAD=0" "SLPX_TWIN=0" "SLPX_TAPE_TEMPLATES=0" "SLPX_MF_SOLVE=0" \
         "SLPX_RESTORATION_PREFETCH=0" "SLPX_SETUP_THREADS=1"; do
  echo -n "$v: "
  env $v timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
done
