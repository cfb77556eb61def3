#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03b; mkdir -p $out
timeout 600 python tools/exp_train.py f16 30 2>&1 | grep exp_train | tee $out/exp_train_f16.txt
timeout 600 python tools/exp_train.py bf16 30 2>&1 | grep exp_train | tee $out/exp_train_bf16.txt
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "trajectory" -s 2>&1 | grep -E "trajectory|passed|failed|Error|assert" | tee $out/traj.txt
