for v in "SLPX_FUSE_LAUNCHES=0" "SLPX_FUSE_KKT=0" "SLPX_FUSE_BACKSUB=0" "SLPX_FUSE_SOLVE=0" "SLPX_SINGLE_LAUNCH=0" "SLPX_LDLT_IL=0" "SLPX_IL_DIRECT=0"; do
  echo -n "$v: "
  env $v timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1
done
