#!/bin/bash
# does the NUMBER of (shared) side streams matter beyond "no second set"?  scoring pass and train step with 1 / 2 / 3 streams in the pool
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for v in default ss2 ss1; do
  if [ $v == default ]; then unset MNF_LIB_PATH; else export MNF_LIB_PATH=$PWD/gpurun_exp/lib_$v.so; fi
  echo "== side streams: $v"
  python tools/exp_score.py 256,32 5 2>&1 | grep "exp_score" | grep score_views
  python tools/exp_train.py f16 40 0 8192 0 2>&1 | grep exp_train
done
} | tee gpurun_out/r03_side_streams.txt
