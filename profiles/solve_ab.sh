# whole solves: twin attempts (SLPX_TWIN), the look-ahead iteration (SLPX_IPM_LOOKAHEAD), new right-hand sides through
# the fronts (SLPX_MF_SOLVE, r05: the iterates differ from the pair lists' in the last bits, and with them the
# iteration counts), the common iteration decided on the device (SLPX_IPM_PIPELINE, r06), the Hessian's rows per stage family
# (SLPX_HESSIAN_FAMILIES, r06: a setup switch — the iterates differ in the last bits) off one at a time, same box
for N in 100 300 500 1000; do
  for V in "" "SLPX_TWIN=0" "SLPX_IPM_LOOKAHEAD=0" "SLPX_MF_SOLVE=0" "SLPX_IPM_PIPELINE=0" "SLPX_IPM_RIDE=0" "SLPX_HESSIAN_FAMILIES=0"; do
    echo "== N=$N ${V:-default}"
    env $V SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | tail -2
  done
done
