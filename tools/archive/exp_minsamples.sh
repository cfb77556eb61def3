#!/bin/bash
# diagnostic: field-kernel ns/column when a round marches 4 / 8 / 16 / 32 samples per ray (MNF_MIN_SAMPLES; not the
# reference's schedule): how much the gather gains when a wave walks further along the same rays
cd "$(dirname "$0")/.."
for w in random trained; do for m in 4 8 16 32 64; do
  echo "== weights $w min_samples $m"
  MNF_MIN_SAMPLES=$m MNF_ROUND_LOG=1 timeout 300 python bench.py --workload render800 --weights $w --views 1 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>&1 >/dev/null | grep "mnf round" | awk '{c+=$5; t+=$7} NR<=2 {print} END {printf "total cols %d field ms %.3f -> %.4f ns/col\n", c, t, t*1e6/c}'
done; done
