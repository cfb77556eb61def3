#!/bin/bash
# tools/r05_dropin_timeline.sh: kernel timeline of one step of the UNCHANGED caller's loop (rocprofv3 kernel trace of exp_dropin_profile.py) -> gpurun_out/r05_dropin_timeline.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_dropin_profile.py 14 short > $GRAFT_REPO_ROOT/gpurun_out/tl/exp.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tl/prof -name "*kernel_trace.csv" | head -1)
python tools/analyze_trace.py $f sample_rays_kernel -3 > gpurun_out/r05_dropin_timeline.txt
rm -rf gpurun_out/tl/prof
tail -2 gpurun_out/tl/exp.txt; cat gpurun_out/r05_dropin_timeline.txt
