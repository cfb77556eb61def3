export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
timeout 300 python tools/debug_bins.py 14 2>&1 | grep -v amdgpu | tail -8; timeout 300 python tools/debug_bins.py 19 200000 2>&1 | grep -v amdgpu | tail -8
mkdir -p gpurun_out/sc
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sc/prof -- python3 $GRAFT_REPO_ROOT/tools/exp_scatter.py > $GRAFT_REPO_ROOT/gpurun_out/sc/exp.txt 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/sc/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/sc/stats.csv \; ; rm -rf gpurun_out/sc/prof; grep -i "bin_\|hash_bwd\|wgrad\|fillBuffer" gpurun_out/sc/stats.csv | cut -c1-160
