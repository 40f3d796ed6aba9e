# whole solves: twin attempts (SLPX_TWIN), the look-ahead iteration (SLPX_IPM_LOOKAHEAD), new right-hand sides through
# the fronts (SLPX_MF_SOLVE, r05: the iterates differ from the pair lists' in the last bits, and with them the
# iteration counts), the restoration system compiled ahead (SLPX_RESTORATION_PREFETCH, r05) off one at a time, same box
for N in 100 300 500; do
  for V in "" "SLPX_TWIN=0" "SLPX_IPM_LOOKAHEAD=0" "SLPX_MF_SOLVE=0" "SLPX_RESTORATION_PREFETCH=0"; do
    echo "== N=$N ${V:-default}"
    env $V SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | tail -2
  done
done
