# interleaved factorization + backward solve with every round in one launch (small batches) against a launch per round
# (SLPX_IL_SINGLE_MAX=0), one box:  bash profiles/il_single_probe.sh
run() {
  env $1 timeout 600 python bench.py --workload batch512 --batch $2 --N $3 --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']), 'steps/s', round(d['ms_per_step'], 4), 'ms/step', {k: round(v, 4) for k, v in d['roofline']['per_kernel_ms'].items()}, 'rounds', d['config']['ldlt_rounds'], 'tasks', d['config']['ldlt_tasks'], 'failed', d['per_problem']['failed'])
"
}
for cfg in "64 500" "128 500" "64 1000" "256 500"; do
  set -- $cfg
  for env in "SLPX_IL_SINGLE_MAX=100000" "SLPX_IL_SINGLE_MAX=0" "SLPX_IL_SINGLE_MAX=100000 SLPX_TASK_ENTRIES=768" "SLPX_IL_SINGLE_MAX=100000 SLPX_TASK_ENTRIES=384"; do
    echo -n "$1 x N=$2 $env: "; run "$env" $1 $2
  done
done
