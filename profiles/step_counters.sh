#!/bin/bash
# Instruction and LDS counters of the default workload's kernels (cart-pole N=1000, one problem), two --pmc passes on
# their own (never with a trace: MI355X_MICROARCH.md): what a launch of the step kernel EXECUTES — vector, scalar and
# LDS instructions per wave, LDS bank conflicts — beside its wave cycles.
#   bash profiles/step_counters.sh [tag]   -> gpurun_out/<tag>_step_counters.json
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r05}
export PYTHONPATH=$R TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
FAST="--no-cpu-baseline --no-batched --no-whole-solve --steps 20 --warmup 2 --repeats 1 --fixed-repeats"
rm -rf $O/pmc_a $O/pmc_b
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_a -- python $R/bench.py $FAST > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_b -- python $R/bench.py $FAST > $O/pmc_b.log 2>&1
python - $O/pmc_a $O/pmc_b > $O/${TAG}_step_counters.json <<'PY'
import collections, csv, glob, json, re, sys
def short(name):
    name = re.sub(r"^void ", "", name); name = re.sub(r"slpx::", "", name); return re.sub(r"\(.*", "", name)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = max(len(v) for v in cs.values())
    w = e.get("SQ_WAVES")
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            if c in e: e[c + "_per_wave"] = e[c] / w
    out[k] = e
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_a $O/pmc_b
python - $O/${TAG}_step_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    if "mf_step" in k or "tape_templates" in k: print(k[:50], {c: round(v, 1) for c, v in e.items()})
PY
