# chains only from the deepest child (SLPX_SN_DEEPEST: 0 off, 1 every task, 2 from round 1 up): the step kernel alone, same box
for D in 2 1 0; do
  echo "== SLPX_SN_DEEPEST=$D"
  SLPX_SN_DEEPEST=$D PYTHONPATH=$PWD python profiles/mf_time.py 1000 500 5000 100 300 gfold
done
