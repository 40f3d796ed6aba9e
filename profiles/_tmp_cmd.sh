cd $GRAFT_REPO_ROOT
for R in 1 2 3; do for L in prev new; do
  cp build/libslpx_$L.so sleipnir_amd/libslpx.so
  echo "== $L"
  python bench.py --steps 2000 --warmup 200 --repeats 3 --no-whole-solve --no-batched --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  for N in 100 500; do PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | grep "^$N" | awk '{print $1,$3,$4,$6}' | tr '\n' ';'; echo; done
done; done
cp build/libslpx_new.so sleipnir_amd/libslpx.so
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3)
for v in "SLPX_IPM_RESIDENT=0" "SLPX_PRELAUNCH=1" "SLPX_IPM_LOOKAHEAD_RIDE=1"; do echo -n "$v: "; env $v timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done > gpurun_out/matrix_rest.txt 2>&1
cat gpurun_out/matrix_rest.txt
