#!/bin/bash
# experiment: time of hash_bwd_kernel per range of levels (results are wrong for the skipped levels; timing only)
export TMPDIR=/tmp
for r in 0,4 4,8 8,12 12,16 0,1 15,16; do
  rm -rf gpurun_out/hbl
  MNF_HASH_BWD_LEVELS=$r rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hbl -- python3 bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  echo "levels $r: $(grep hash_bwd gpurun_out/hbl/*/*kernel_stats.csv | cut -d, -f1-4 | tail -c 60)"
done
