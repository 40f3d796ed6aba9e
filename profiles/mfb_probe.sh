set -x

SLPX_MF_BATCH=1 timeout 900 python -m pytest tests/test_timed_path_parity_gpu.py tests/test_gpu_parity.py tests/test_parity_holes_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -3
for b in "64 500" "512 1000"; do
  set -- $b
  for env in "SLPX_MF_BATCH=0" "SLPX_MF_BATCH=1" "SLPX_MF_BATCH=1 SLPX_MFB_THREADS=512" "SLPX_MF_BATCH=1 SLPX_MFB_THREADS=1024" "SLPX_MF_BATCH=1 SLPX_IL_MIN_BATCH=100000"; do
    echo "== batch $1 N $2 $env"
    env $env timeout 300 python bench.py --workload batch512 --batch $1 --N $2 --steps 30 --warmup 5 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d.get('ms_per_ldlt_factor'), d.get('ms_per_ldlt_solve'), d['per_problem']['failed'])
    else: print(line[-300:])
"
  done
done
