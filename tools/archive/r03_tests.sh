#!/bin/bash
# GPU test suite (stop at first failure) -> gpurun_out/r03_tests.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40 > gpurun_out/r03_tests.txt
cat gpurun_out/r03_tests.txt
