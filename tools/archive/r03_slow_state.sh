#!/bin/bash
# kernel timeline of a train step in the slow state (after 300 steps of a random-init field in the same process)
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
python tools/exp_sustained.py 1 > /dev/null 2>&1
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_after_training3.py early > $GRAFT_REPO_ROOT/gpurun_out/tl/exp.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tl/prof -name "*kernel_trace.csv" | head -1)
grep "ms/step" gpurun_out/tl/exp.txt
python tools/analyze_trace.py $f planes_kernel -3 > gpurun_out/r03_slow_state_timeline.txt
python tools/analyze_trace.py $f planes_kernel 20 > gpurun_out/r03_fast_state_timeline.txt
rm -rf gpurun_out/tl/prof
echo "=== fast state (before)"; grep -v "^ .* +  *[0-9]\.[0-9]  gap" gpurun_out/r03_fast_state_timeline.txt | head -40
echo "=== slow state (after)"; grep -v "^ .* +  *[0-9]\.[0-9]  gap" gpurun_out/r03_slow_state_timeline.txt | head -40
