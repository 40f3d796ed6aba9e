# widest supernode / balanced cuts of over-long chains (SLPX_SN_MAX_WIDTH, SLPX_SN_BALANCE): the step kernel alone, same box
for W in 8 7 6 5; do for B in 1 0; do
  echo "== SLPX_SN_MAX_WIDTH=$W SLPX_SN_BALANCE=$B"
  SLPX_SN_MAX_WIDTH=$W SLPX_SN_BALANCE=$B PYTHONPATH=$PWD python profiles/mf_time.py 1000 500 5000 gfold
done; done
