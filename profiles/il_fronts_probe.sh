# batch factorization by fronts with four lanes per problem (ldlt_mfq_kernels.h) against the pair-list kernel
# (SLPX_IL_FRONTS=0), and the task size under it — one box:  bash profiles/il_fronts_probe.sh
run() {
  env $1 timeout 600 python bench.py --workload batch512 --batch $2 --N $3 --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']), 'steps/s', round(d['ms_per_step'], 4), 'ms/step', {k: round(v, 4) for k, v in d['roofline']['per_kernel_ms'].items()}, 'rounds', d['config']['ldlt_rounds'], 'tasks', d['config']['ldlt_tasks'], 'nnzL', d['config']['nnz_L'], 'failed', d['per_problem']['failed'])
"
}
for cfg in "64 500" "512 500" "512 1000"; do
  set -- $cfg
  for env in "A=0" "SLPX_IL_FRONTS=0" "SLPX_TASK_ENTRIES=256" "SLPX_TASK_ENTRIES=384" "SLPX_TASK_ENTRIES=512" "SLPX_TASK_ENTRIES=640"; do
    echo -n "$1 x N=$2 $env: "; run "$env" $1 $2
  done
done
