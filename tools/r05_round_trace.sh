#!/bin/bash
# tools/r05_round_trace.sh [views] [score]: rocprofv3 kernel trace of ONE render job (tools/exp_round_log.py with the product library) -> per-round march / field durations
# and the gaps between them -> gpurun_out/r05_round_trace_<tag>.txt
export TMPDIR=/tmp
tag=${2:-render}; mkdir -p gpurun_out/rt
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rt/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_round_log.py ${1:-2} ${2:-} > $GRAFT_REPO_ROOT/gpurun_out/rt/exp.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/rt/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/r05_round_trace_$tag.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
last_init = max(i for i, e in enumerate(ev) if "init_kernel" in e[2])
ev = ev[last_init:]
rounds = []
for i, e in enumerate(ev):
    if "round_march" in e[2] and i + 1 < len(ev) and "field_kernel" in ev[i + 1][2]:
        f = ev[i + 1]
        prev_end = ev[i - 1][1]
        rounds.append(((e[1] - e[0]) / 1e3, (f[1] - f[0]) / 1e3, (e[0] - prev_end) / 1e3, (f[0] - e[1]) / 1e3))
tot = (ev[-1][1] - ev[0][0]) / 1e3
print(f"{len(rounds)} rounds; call {tot:.0f} us from init to the last kernel; march {sum(r[0] for r in rounds):.0f} us, field {sum(r[1] for r in rounds):.0f} us, "
      f"gaps in front of march {sum(max(r[2], 0) for r in rounds):.0f} us, between march and field {sum(max(r[3], 0) for r in rounds):.0f} us")
for k, r in enumerate(rounds):
    if k % 4 == 0: print(f"round {k}: march {r[0]:.1f}  field {r[1]:.1f}  gap before {r[2]:.1f}  gap between {r[3]:.1f}")
PY
rm -rf gpurun_out/rt/prof
cat gpurun_out/rt/exp.txt | tail -1; head -50 gpurun_out/r05_round_trace_$tag.txt
