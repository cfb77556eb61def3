#!/bin/bash
# per-kernel times of the scatter under library variants gpurun_exp/lib_<name>_diag.so
export TMPDIR=/tmp
mkdir -p gpurun_out/sc
for v in "$@"; do
  export MNF_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/lib_${v}_diag.so
  [ "$v" == "default" ] && export MNF_LIB_PATH=$GRAFT_REPO_ROOT/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sc/prof_$v -- python3 $GRAFT_REPO_ROOT/tools/exp_scatter.py > $GRAFT_REPO_ROOT/gpurun_out/sc/exp_$v.txt 2>&1
  cd $GRAFT_REPO_ROOT; find gpurun_out/sc/prof_$v -name "*kernel_stats.csv" -exec cp {} gpurun_out/sc/stats_$v.csv \; ; rm -rf gpurun_out/sc/prof_$v
  echo "== $v"; tail -1 gpurun_out/sc/exp_$v.txt; grep -i "bin_\|hash_bwd_walk_kernel<false>\|wgrad_kernel" gpurun_out/sc/stats_$v.csv | cut -d, -f1-4 | cut -c1-150
done
