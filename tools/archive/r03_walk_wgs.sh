#!/bin/bash
# step time (8192 rays, asynchronous, lr 0) by the number of persistent walk workgroups per CU (0: one workgroup per 1024 samples and level).
# profiles/r03_bins_schedule_experiments.txt also holds the two dropped schedules of the binned passes (level by level, half by half).
export TMPDIR=/tmp
export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
mkdir -p gpurun_out
{
for rep in 1 2; do
for w in 0 4 8 16; do
echo "== walk workgroups per CU $w"; MNF_WALK_WGS=$w python tools/exp_train.py f16 40 0 8192 0 2>&1 | grep exp_train
done
done
} | tee gpurun_out/r03_walk_wgs.txt
