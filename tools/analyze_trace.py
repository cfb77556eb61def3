"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration, stream/queue, and the
idle time of the GPU between launches.  usage: python tools/analyze_trace.py <kernel_trace.csv> [anchor-kernel-substring] [which occurrence]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "planes_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("anchor kernel not found twice")
a, b = idx[which], idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
busy_until = t0
idle = 0
print(f"step of {b - a} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us anchor to anchor")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_until
    if gap > 0:
        idle += gap
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mnf::", "").replace("(anonymous namespace)::", "")[:48]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {max(gap, 0) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    busy_until = max(busy_until, e)
print(f"GPU idle inside the step: {idle / 1e3:.1f} us")
