#!/bin/bash
# tools/r05_prep_wait.sh: kernel trace of a short 800x800 render bench (several render jobs in flight) -> tools/analyze_prep_wait.py -> gpurun_out/r05_prep_wait.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/pw
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pw/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload render800 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-views1 > $GRAFT_REPO_ROOT/gpurun_out/pw/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/pw/bench.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/pw/prof -name "*kernel_trace.csv" | head -1)
python tools/analyze_prep_wait.py $f > gpurun_out/r05_prep_wait.txt 2>&1
rm -rf gpurun_out/pw/prof
cat gpurun_out/r05_prep_wait.txt
