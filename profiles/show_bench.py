#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: python profiles/show_bench.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms/step", d["timing"]["block_ms"])
r = d["roofline"]
print({k: r.get(k) for k in ("kernel", "achieved", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch",
                             "step_launches_ms", "per_kernel_ms")})
print("setup", d.get("setup_s"))
print("whole solve", d.get("whole_solve"))
print("cpu", d.get("cpu_baseline"))
for b in d.get("batched", []):
    print(b["workload"], b["steps_per_s"], b["per_kernel_ms"], b["hbm_frac"], b.get("traffic"))
