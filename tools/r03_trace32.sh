#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/trace32; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/exp_score.py 32 2 > $out/log.txt 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $out/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last pass only: take the last 1200 launches of the three round kernels
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("round_prep", "round_march", "field_kernel<128, 2, 2"))]
sel = sel[-2 * 3 * 152:]
import collections
d = collections.defaultdict(list)
for r in sel:
    k = "prep" if "round_prep" in r["Kernel_Name"] else ("march" if "round_march" in r["Kernel_Name"] else "field")
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v2 = sorted(v)
    print(k, "n", len(v), "sum %.1f us" % sum(v), "median %.1f" % v2[len(v2) // 2], "p10 %.1f" % v2[len(v2) // 10], "p90 %.1f" % v2[9 * len(v2) // 10], "min %.1f" % v2[0], "max %.1f" % v2[-1])
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print("span of these launches: %.1f us" % ((t1 - t0) / 1e3))
# per queue timeline sample: first 24 launches
for r in sel[300:336]:
    print(r["Queue_Id"], "%.1f" % ((int(r["Start_Timestamp"]) - t0) / 1e3), "+%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), r["Kernel_Name"].split("(")[0][-40:])
PY
rm -rf $out/t; cat $out/summary.txt
