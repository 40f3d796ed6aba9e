# whole solves: twin attempts (SLPX_TWIN), the look-ahead iteration (SLPX_IPM_LOOKAHEAD)
# off one at a time, same box
for N in 100 300 500; do
  for V in "" "SLPX_TWIN=0" "SLPX_IPM_LOOKAHEAD=0"; do
    echo "== N=$N ${V:-default}"
    env $V SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | tail -2
  done
done
