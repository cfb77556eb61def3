#!/bin/bash
# more than two render jobs in flight, with enough hardware queues for them?  (ROCm multiplexes streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== 3 shared side streams (release), default hardware queues"
python tools/exp_split.py 2>&1 | grep "exp_split" | grep -v "8x8"
echo "== 7 shared side streams, default hardware queues"
MNF_LIB_PATH=$PWD/gpurun_exp/lib_ss7.so python tools/exp_split.py 2>&1 | grep "exp_split" | grep -v "8x8"
echo "== 7 shared side streams, GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 MNF_LIB_PATH=$PWD/gpurun_exp/lib_ss7.so python tools/exp_split.py 2>&1 | grep "exp_split" | grep -v "8x8"
echo "== 3 shared side streams (release), GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 python tools/exp_split.py 2>&1 | grep "exp_split" | grep -v "8x8"
} | tee gpurun_out/r03_hw_queues.txt
