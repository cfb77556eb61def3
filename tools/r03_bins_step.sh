#!/bin/bash
# train-step time (8192 rays, asynchronous, parameters frozen: lr 0) by first binned level, bins on their own stream or in front of the walk
export TMPDIR=/tmp
export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
mkdir -p gpurun_out
{
for b in 16 13 12 11 10 8; do
echo "== first binned level $b, own stream";  MNF_BIN_LEVEL0=$b python tools/exp_train.py f16 40 0 8192 0 2>&1 | grep exp_train
echo "== first binned level $b, same stream"; MNF_BIN_SAME_STREAM=1 MNF_BIN_LEVEL0=$b python tools/exp_train.py f16 40 0 8192 0 2>&1 | grep exp_train
done
} | tee gpurun_out/r03_bins_step.txt
