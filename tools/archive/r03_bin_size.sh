#!/bin/bash
# bins of 4096 entries (128 KB of LDS doubles, 1024 threads, one pass-B workgroup per CU) against 2048 entries (64 KB, 512 threads, two per CU)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in bin12 bin11; do
  echo "== $v"; MNF_LIB_PATH=$PWD/gpurun_exp/lib_$v.so python tools/exp_train.py f16 40 0 8192 0 2>&1 | grep exp_train
  TAG="$v alone" MNF_NO_WGRAD=1 MNF_LIB_PATH=$PWD/gpurun_exp/lib_${v}_diag.so python tools/exp_scatter.py 2>&1 | grep kept
done
done
} | tee gpurun_out/r03_bin_size.txt
