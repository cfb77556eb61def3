#!/bin/bash
# tools/exp_knock.sh <name>...  per-round field-kernel time of knock-out builds (gpurun_exp/lib_<name>.so, see MNF_KNOCK in csrc/field.hip):
# rounds 0-3 of one 800x800 view march the same 2.56 M columns in every build, so their ns/column are comparable.
cd "$(dirname "$0")/.."
for name in "$@"; do
  echo "== $name"
  MNF_ROUND_LOG=1 MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 300 python bench.py --workload render800 --weights random --views 1 --steps 2 --warmup 0 --no-cpu-baseline --no-kernel-timing 2>&1 >/dev/null | grep "mnf round [0-3]\]" | tail -4
done
