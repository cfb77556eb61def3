#!/bin/bash
# scatter timing (diag library): alone on the chip and beside wgrad; MNF_BIN_LEVEL0 = first level that goes through the bins (16: none)
export TMPDIR=/tmp
PKG=active-perception-using-neural-radiance-fields_amd
export MNF_LIB_PATH=$PWD/$PKG/libmi355nerf_diag.so
mkdir -p gpurun_out
{
for b in 16 12 10 8; do
TAG="bins from level $b, beside wgrad " MNF_BIN_LEVEL0=$b python tools/exp_scatter.py
TAG="bins from level $b, alone        " MNF_BIN_LEVEL0=$b MNF_NO_WGRAD=1 python tools/exp_scatter.py
done
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r03_scatter_exp.txt
