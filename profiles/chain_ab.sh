# chained steps (DeviceNlp::sweep_full_for_step; the default where the step kernel leaves the sweep room: not at
# N=5000) against the sweep in the main stream (SLPX_CHAIN_TAPE=0), same box:
#   bash profiles/chain_ab.sh > gpurun_out/chain_ab.txt
for N in ${CHAIN_AB_N:-1000 5000 300 100}; do
  for v in "SLPX_CHAIN_TAPE=0" "SLPX_CHAIN_TAPE=1" "SLPX_CHAIN_TAPE=0" "SLPX_CHAIN_TAPE=1"; do
    echo -n "N=$N $v: "
    env $v timeout 300 python bench.py --N $N --steps 2000 --warmup 200 --no-cpu-baseline --no-batched --no-whole-solve 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value'], 1), 'steps/s', round(1e3 * d['ms_per_step'], 2), 'us/step; kernels', d['roofline'].get('step_launches_ms'))
"
  done
done
