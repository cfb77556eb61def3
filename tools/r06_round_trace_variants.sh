#!/bin/bash
# tools/r06_round_trace_variants.sh <views> <score|render> <name>...: per-round marcher / field durations of ONE render job (rocprofv3 kernel trace of tools/exp_round_log.py)
# for experiment builds gpurun_exp/lib_<name>.so -> gpurun_out/r06_round_trace_<mode>_<name>.txt
export TMPDIR=/tmp
V=$1; MODE=$2; shift; shift
for name in "$@"; do
  export MNF_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/lib_$name.so
  python3 tools/exp_round_log.py $V $([ "$MODE" = score ] && echo score) > /dev/null 2>&1      # trains / caches the stand-in for this library
  rm -rf /tmp/rt_$name
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rt_$name -- python3 $GRAFT_REPO_ROOT/tools/exp_round_log.py $V $([ "$MODE" = score ] && echo score) > /tmp/rt_$name.txt 2>&1)
  f=$(find /tmp/rt_$name -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > gpurun_out/r06_round_trace_${MODE}_$name.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
last_init = max(i for i, e in enumerate(ev) if "init_kernel" in e[2])
ev = ev[last_init:]
rounds = []
for i, e in enumerate(ev):
    if "round_march" in e[2] and i + 1 < len(ev) and "field_kernel" in ev[i + 1][2]:
        f = ev[i + 1]
        rounds.append(((e[1] - e[0]) / 1e3, (f[1] - f[0]) / 1e3))
tot = (ev[-1][1] - ev[0][0]) / 1e3
print(f"{len(rounds)} rounds; call {tot:.0f} us; march {sum(r[0] for r in rounds):.0f} us, field {sum(r[1] for r in rounds):.0f} us")
print("march by round (every 4th): " + " ".join(f"{r[0]:.0f}" for r in rounds[::4]))
PY
  echo "$name: $(head -1 gpurun_out/r06_round_trace_${MODE}_$name.txt)"; sed -n 2p gpurun_out/r06_round_trace_${MODE}_$name.txt
done
