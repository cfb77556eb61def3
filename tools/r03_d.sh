#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03d; rm -rf $out; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/atomic_bench tools/atomic_bench.hip && /tmp/atomic_bench | tee $out/atomic_bench.txt
MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so MNF_SCATTER_COUNT=1 timeout 600 python tools/exp_train.py f16 2 0 8192 1 2>&1 | grep -E "exp_train|mnf scatter" | tail -4 | tee $out/scatter_count.txt
