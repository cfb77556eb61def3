#!/bin/bash
# tools/r05_score_trace.sh [views]: rocprofv3 kernel trace of scoring passes (tools/exp_score.py) -> the LAST score_views pass: span, GPU idle inside it, time per kernel name
export TMPDIR=/tmp
mkdir -p gpurun_out/st
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_score.py ${1:-32} 3 > $GRAFT_REPO_ROOT/gpurun_out/st/exp.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/st/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/r05_score_trace_${1:-32}.txt <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r.get("Queue_Id", "?")) for r in rows]
sc = [i for i, e in enumerate(ev) if "score_kernel" in e[2]]
# passes of score_views end with score_kernel; score_poses passes too: take the pass that ends at the 4th score kernel from the end (the last score_views pass: exp_score runs views then poses)
end_i = sc[-5] if len(sc) >= 5 else sc[-1]
start_i = sc[-6] + 1 if len(sc) >= 6 else 0
seg = ev[start_i:end_i + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy_until, idle, gaps = t0, 0, []
for s, e, n, q in seg:
    if s > busy_until:
        idle += s - busy_until; gaps.append(((s - busy_until) / 1e3, (s - t0) / 1e3, n))
    busy_until = max(busy_until, e)
per = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in seg:
    per[n][0] += 1; per[n][1] += (e - s) / 1e3
print(f"pass of {len(seg)} launches: {(t1 - t0) / 1e3:.0f} us first kernel start to score_kernel end; GPU idle inside {idle / 1e3:.0f} us")
for n, (c, t) in sorted(per.items(), key=lambda x: -x[1][1])[:14]: print(f"  {t:9.1f} us  {c:5d} x  {n}")
print("largest gaps (us, at us, before kernel):", [(round(g, 1), round(a), n[:30]) for g, a, n in sorted(gaps, reverse=True)[:8]])
print("first 14 launches:"); [print(f"   {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} q{q} {n}") for s, e, n, q in seg[:14]]
print("last 10 launches:"); [print(f"   {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} q{q} {n}") for s, e, n, q in seg[-10:]]
PY
rm -rf gpurun_out/st/prof
tail -4 gpurun_out/st/exp.txt; cat gpurun_out/r05_score_trace_${1:-32}.txt
