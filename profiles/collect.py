#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small files committed under profiles/.

    python profiles/collect.py <round-tag> <stats-dir> <pmc-fetch-dir> <pmc-write-dir>

Inputs (made on the GPU box, see profiles/README.md for the exact commands):
  stats-dir      rocprofv3 --kernel-trace --stats --output-format csv  -- python bench.py
  pmc-fetch-dir  rocprofv3 --pmc FETCH_SIZE  --output-format csv       -- python bench.py --steps 20 ...
  pmc-write-dir  rocprofv3 --pmc WRITE_SIZE  --output-format csv       -- python bench.py --steps 20 ...
(counters in their own passes, never together with a trace: MI355X_MICROARCH.md "rocprofv3 PMC
slots": FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2.)

Outputs:
  profiles/<tag>_kernel_stats.csv   the --stats summary, verbatim
  profiles/<tag>_traffic.json       per kernel and grid size: launches, mean FETCH_SIZE and
                                    WRITE_SIZE in KB as reported, and HBM bytes per launch
                                    with the guide's gfx950 correction (FETCH_SIZE counts
                                    128-B requests as 64 B for wide coalesced reads: x2;
                                    WRITE_SIZE is uncalibrated and used as reported)
bench.py reads <tag>_traffic.json to fill roofline.traffic.
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"slpx::", "", name)
    return re.sub(r"\(.*", "", name)


def counters(directory: str, counter: str):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    stats = glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True)
    if stats:
        shutil.copy(stats[0], HERE / f"{tag}_kernel_stats.csv")
    fetch, write = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    table = {}
    for key in sorted(set(fetch) | set(write)):
        kernel, grid = key
        f, w = fetch.get(key, []), write.get(key, [])
        fkb = sum(f) / len(f) if f else None
        wkb = sum(w) / len(w) if w else None
        table.setdefault(kernel, {})[str(grid)] = {
            "launches_sampled": max(len(f), len(w)),
            "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
            "hbm_bytes_per_launch": (2.0 * 1024 * fkb if fkb is not None else 0.0)
                                    + (1024 * wkb if wkb is not None else 0.0),
        }
    (HERE / f"{tag}_traffic.json").write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
    print(f"wrote profiles/{tag}_kernel_stats.csv and profiles/{tag}_traffic.json ({len(table)} kernels)")


if __name__ == "__main__":
    main()
