#!/usr/bin/env python3
"""Prints the kernel timeline of two consecutive Newton steps from a rocprofv3
--kernel-trace CSV:  python profiles/timeline.py <kernel_trace.csv> [step-index]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"slpx::", "", name)
    return re.sub(r"\(.*", "", name)[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"])
                for r in rows)
    # a step starts with the tape's generated kernel (older profiles: the 64-thread interpreter)
    first = "slpx_tape_templates" if any(e[2].startswith("slpx_tape_templates") for e in ev) else "tape_sweep_lds_kernel<64"
    starts = [i for i, e in enumerate(ev) if e[2].startswith(first)]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
    i0, i1 = starts[k], starts[k + 2]
    t0 = ev[i0][0]
    print("start_us  duration_us  queue  kernel")
    for s, e, n, q in ev[i0:i1 + 1]:
        print(f"{(s - t0) / 1000:9.2f}  {(e - s) / 1000:9.2f}  q{q}  {n}")


if __name__ == "__main__":
    main()
