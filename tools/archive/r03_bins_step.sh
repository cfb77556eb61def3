#!/bin/bash
# train-step time (asynchronous, parameters frozen: lr 0) by first binned level; $1 = rays per batch (8192: BASELINE config 5, 2000: the reference yaml)
export TMPDIR=/tmp
export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
mkdir -p gpurun_out
R=${1:-8192}
{
for b in 16 13 12 11 10 8; do
echo "== rays $R, first binned level $b";  MNF_BIN_LEVEL0=$b python tools/exp_train.py f16 40 0 $R 0 2>&1 | grep exp_train
done
} | tee gpurun_out/r03_bins_step_$R.txt
