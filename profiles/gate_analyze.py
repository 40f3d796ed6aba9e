import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
gaps, durs, nxt = [], [], []
for i, (s, e, n) in enumerate(ev):
    if "ldlt_mf_twin_kernel" in n or "ldlt_mf_step_kernel" in n:
        if i > 0:
            gaps.append((s - ev[i-1][1]) / 1e3)
            durs.append((e - s) / 1e3)
        if i + 1 < len(ev): nxt.append((ev[i+1][0] - e) / 1e3)
def q(v): 
    v = sorted(v); return f"n={len(v)} median {v[len(v)//2]:.2f} mean {sum(v)/len(v):.2f} p10 {v[len(v)//10]:.2f} p90 {v[len(v)*9//10]:.2f}"
print("gap before step kernel:", q(gaps)); print("step kernel duration:  ", q(durs)); print("gap after step kernel: ", q(nxt))
print("total span ms", (ev[-1][1]-ev[0][0])/1e6, "kernels", len(ev))
