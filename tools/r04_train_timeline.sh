#!/bin/bash
# tools/r04_train_timeline.sh [rays] [backward mode] [presample 0|1]: kernel timeline of one asynchronous train step (rocprofv3 kernel trace of exp_train.py) -> gpurun_out/r04_train_timeline_${1:-8192}_mode${2:-0}[_presampled].txt
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_train.py f16 12 0 ${1:-8192} 0 ${2:-0} ${3:-0} > $GRAFT_REPO_ROOT/gpurun_out/tl/exp.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tl/prof -name "*kernel_trace.csv" | head -1)
sfx=""; anchor=planes_kernel; if [ "${3:-0}" = 1 ]; then sfx=_presampled; anchor=gather_fragsT_kernel; fi
python tools/analyze_trace.py $f $anchor -3 > gpurun_out/r04_train_timeline_${1:-8192}_mode${2:-0}$sfx.txt
rm -rf gpurun_out/tl/prof
tail -2 gpurun_out/tl/exp.txt; cat gpurun_out/r04_train_timeline_${1:-8192}_mode${2:-0}$sfx.txt
