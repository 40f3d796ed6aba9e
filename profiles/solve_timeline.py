"""Launch timeline of the last whole solve in a rocprofv3 kernel trace: start offset, duration (us), name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def window(tail, title):
    print(f"# {title}")
    t0 = int(tail[0]["Start_Timestamp"])
    prev_end = t0
    for r in tail:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - t0) / 1e3:9.2f}  gap {(st - prev_end) / 1e3:6.2f}  dur {(en - st) / 1e3:7.2f}  {r['Kernel_Name'][:60]}")
        prev_end = en


window(rows[-500:-380], "launches 500 .. 380 from the end of the last solve")
# (r06) a stretch of OUTER iterations (the look-ahead chain of ipm_core_resident) and one of RESTORATION iterations
# (restoration.hip's kernels), wherever the last ones of each are
for name, title in (("ipm_lookahead_kernel", "outer iterations"), ("fr_expand_kernel", "restoration iterations")):
    hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
    if len(hits) > 40:
        i = hits[-30]
        window(rows[max(0, i - 10):i + 60], f"{title}: around one of the last {name} launches")
