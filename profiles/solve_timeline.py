"""Launch timeline of the last whole solve in a rocprofv3 kernel trace: start offset, duration (us), name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-500:-380]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = t0
for r in tail:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:9.2f}  gap {(st - prev_end) / 1e3:6.2f}  dur {(en - st) / 1e3:7.2f}  {r['Kernel_Name'][:60]}")
    prev_end = en
