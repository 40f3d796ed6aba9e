#!/bin/bash
# How much of the host's reaction between two chained steps is hidden under the step kernel's tail: the host sits on
# every step's verdict for D nanoseconds (SLPX_DEBUG_STATS_DELAY_NS) before it launches the next step.
#   bash profiles/host_slack.sh [repeats]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
REP=${1:-2}
export PYTHONPATH=$R
for i in $(seq $REP); do
  for D in 0 500 1000 2000 3000 4000 6000 8000; do
    SLPX_DEBUG_STATS_DELAY_NS=$D python $R/bench.py --steps 500 --warmup 50 2>/dev/null | grep "^{" | tail -1 > /tmp/hs.json
    python - $D <<'PY'
import json, sys
d = json.load(open("/tmp/hs.json"))
print(f"host delay {int(sys.argv[1]):5d} ns: {1e3 * d['ms_per_step']:.2f} us/step")
PY
  done
done
