#!/bin/bash
# The bench line of two builds of the library on the same box, alternating (box-to-box differences are as large as
# the changes being measured): build/base_lib/libslpx.so (a build of an earlier commit) against the tree's.
#   bash profiles/lib_ab.sh [repeats] [bench args ...]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
REP=${1:-3}
shift || true
export PYTHONPATH=$R
# (the test-support libraries link the libslpx they were built beside: they are rebuilt for the library under test
# and removed afterwards, so that the next user builds them against the tree's again)
trap 'rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so' EXIT
rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
mkdir -p $R/gpurun_out
for i in $(seq $REP); do
  for which in base tree; do
    if [ $which = base ]; then D=$R/build/base_lib; else D=$R/sleipnir_amd; fi
    rm -f $R/tests/support/libslpx_models.so
    SLPX_LIB=$D/libslpx.so LD_LIBRARY_PATH=$D:${LD_LIBRARY_PATH:-} python $R/bench.py "$@" 2>/dev/null | grep "^{" | tail -1 > $R/gpurun_out/ab_$which.json
    python - "$which" $R/gpurun_out/ab_$which.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
print(f"{sys.argv[1]:5s} {d['value']:9.1f} steps/s  {1e3 * d['ms_per_step']:.2f} us/step   step kernel {1e3 * r['launch_ms']:.2f} us   sweep {1e3 * r['step_launches_ms']['tape_sweep']:.2f} us")
PY
  done
done
