#!/bin/bash
# tools/r06_small_batch_knobs.sh: the backward's schedule knobs at the reference yaml's 2000 rays (they were tuned at 8192): tile ranges of dgrad, first binned level, wgrad split
# (diagnostic library) -> gpurun_out/r06_small_batch_knobs.txt
export TMPDIR=/tmp
export MNF_LIB_PATH=$GRAFT_REPO_ROOT/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
python3 tools/exp_train.py f16 2 0 2000 0 > /dev/null 2>&1
for spec in none MNF_BWD_CHUNKS=1 MNF_BWD_CHUNKS=2 MNF_BWD_CHUNKS=3 MNF_BIN_LEVEL0=16 MNF_BIN_LEVEL0=14 MNF_BIN_LEVEL0=13 MNF_BIN_LEVEL0=11 MNF_WGRAD_SPLIT=100 MNF_WGRAD_SPLIT=200 MNF_WGRAD_SPLIT=500 MNF_WALK_WGS=4 MNF_WALK_WGS=16 none; do
  if [ "$spec" != "none" ]; then export "$spec"; fi
  echo "== $spec: $(python3 tools/exp_train.py f16 60 0 2000,8192 0 2>&1 | grep 'exp_train\] f16' | sed 's/.*presample=False: //' | tr '\n' '|')"
  if [ "$spec" != "none" ]; then unset "${spec%%=*}"; fi
done
