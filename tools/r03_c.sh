#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03c; rm -rf $out; mkdir -p $out
timeout 600 python tools/exp_train.py f16 30 0 8192,2000 1,0 2>&1 | grep exp_train | tee $out/exp_train_f16.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/exp_train.py f16 12 0 8192 0 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/analyze_trace.py $f planes_kernel -3 > $out/timeline_async.txt
cp $f $out/kernel_trace_async.csv; rm -rf $out/trace
head -80 $out/timeline_async.txt
