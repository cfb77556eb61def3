"""Why `round_prep_kernel` (a one-workgroup kernel) shows 100+ us in a trace with several render jobs in flight: for every prep dispatch of a rocprofv3 --kernel-trace CSV,
the field kernel of ANOTHER queue that was running when the prep was dispatched, and how far that kernel's end is from the prep's end.
    python tools/analyze_prep_wait.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows]
field = sorted([e for e in ev if "field_kernel" in e[3]])
preps = [e for e in ev if "round_prep" in e[3]]
marches = [e for e in ev if "round_march" in e[3]]
fs = np.array([e[0] for e in field]); fe = np.array([e[1] for e in field])
for name, ks in (("round_prep_kernel", preps), ("round_march_kernel", marches)):
    dur = np.array([(e[1] - e[0]) / 1e3 for e in ks])
    behind, slack = [], []
    for s, e, q, _ in ks:
        run = [(fe[i], field[i][2]) for i in np.nonzero((fs <= s) & (fe > s))[0] if field[i][2] != q]
        if run:
            behind.append((e - s) / 1e3)
            slack.append((e - max(x[0] for x in run)) / 1e3)      # > 0: the small kernel ended that long AFTER the other job's field kernel
    print(f"{name}: {len(ks)} dispatches, duration avg {dur.mean():.1f} us, median {np.median(dur):.1f}, max {dur.max():.1f}; "
          f"{len(behind)} dispatched while another job's field kernel held the compute units: avg duration {np.mean(behind) if behind else 0:.1f} us, "
          f"end minus that field kernel's end: median {np.median(slack) if slack else 0:.1f} us (5 % {np.percentile(slack, 5) if slack else 0:.1f}, 95 % {np.percentile(slack, 95) if slack else 0:.1f}); "
          f"the other {len(ks) - len(behind)}: avg {np.mean([d for d, (s, e, q, _) in zip(dur, ks) if not any((fs <= s) & (fe > s))] or [0]):.1f} us")
