# config 4's per-GPU share (64 x N=500) under the batch switches, one box (default: lane-per-problem kernels on
# 512-entry tasks; SLPX_LDLT_IL=0: the per-task pair-list kernels, 1024-entry tasks):
#   bash profiles/b64_probe.sh > gpurun_out/b64_probe.txt
for env in "A=0" "SLPX_TASK_ENTRIES=256" "SLPX_TASK_ENTRIES=384" "SLPX_TASK_ENTRIES=768" "SLPX_LDLT_IL=0" "SLPX_LDLT_IL=0 SLPX_TASK_ENTRIES=2048" "SLPX_LDLT_IL=0 SLPX_SINGLE_LAUNCH=1"; do
  echo -n "64 x N=500 $env: "
  env $env timeout 300 python bench.py --workload batch512 --batch 64 --N 500 --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']), 'steps/s', round(d['ms_per_step'], 4), 'ms/step', {k: round(v, 4) for k, v in d['roofline']['per_kernel_ms'].items()}, 'rounds', d['config']['ldlt_rounds'], 'tasks', d['config']['ldlt_tasks'], 'failed', d['per_problem']['failed'])
"
done
