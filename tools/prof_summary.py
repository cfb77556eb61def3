"""Print a rocprofv3 kernel_stats.csv as per-step milliseconds.  usage: prof_summary.py <kernel_stats.csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6:.1f} ms = {tot/1e6/steps:.2f} ms/step over {steps} steps")
for r in rows[:top]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
