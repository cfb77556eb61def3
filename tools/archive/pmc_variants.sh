#!/bin/bash
# tools/pmc_variants.sh <name>...: dynamic instruction counts of the render field kernel in each gpurun_exp/lib_<name>.so
# (per 64-sample tile = per wave-level gather group: VMEM_RD instructions / 143 in the full kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for name in "$@"; do
  echo "== $name"
  MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 300 bash tools/pmc.sh v_$name SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR 2>&1 | grep "128, 2, 2"
  rm -rf gpurun_out/pmc_v_$name
done
