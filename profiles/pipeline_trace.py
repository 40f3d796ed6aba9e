"""The launches of the last whole solve in a rocprofv3 kernel trace, one line per Newton step launch: for every launch
its tag (T step with two attempts, L look-ahead, S tape sweep, E error norms / decisions, V solve with a new right-hand
side, o others), the gap to the launch before it, its duration (microseconds) and its queue (profiles/pipeline_trace.sh)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-2500:]
t0 = int(tail[0]["Start_Timestamp"]); t1 = int(tail[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
print(f"span {(t1-t0)/1e3:.1f} us busy {busy/1e3:.1f} us frac {busy/(t1-t0):.3f}")
short = {"ldlt_mf_twin": "T", "slpx_tape": "S", "ipm_error": "E", "ipm_lookahead": "L", "ldlt_mf_solve": "V"}
prev_end = t0
line = []
for r in tail:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"]
    tag = next((v for s, v in short.items() if s in k), "o")
    if tag == "T":
        print(" ".join(line)); line = []
    line.append(f"{tag}{(st-prev_end)/1e3:.1f}+{(en-st)/1e3:.1f}q{r.get('Queue_Id', '?')}")
    prev_end = en
