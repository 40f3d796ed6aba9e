#!/usr/bin/env python3
"""Condenses a `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES`
pass (profiles/collect_all.sh, step 6) into one JSON object: per kernel the mean counter values
per launch and, for the kernels that issue matrix instructions, MFMA busy cycles / SQ busy cycles.

    python profiles/mfma_counters.py <pmc-dir>
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"slpx::", "", name)
    return re.sub(r"\(.*", "", name)


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = max(len(v) for v in cs.values())
    if e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / e["SQ_BUSY_CYCLES"]
    out[k] = e
print(json.dumps(out, indent=1))
