#!/bin/bash
export TMPDIR=/tmp
DIAG=$GRAFT_REPO_ROOT/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
export MNF_LIB_PATH=$DIAG
python3 tools/exp_train.py f16 2 0 8192 0 > /dev/null 2>&1
for spec in "none" "MNF_HASH_BWD_LEVELS=0,12" "MNF_HASH_BWD_LEVELS=0,8" "MNF_HASH_BWD_LEVELS=0,5" "none"; do
  if [ "$spec" != "none" ]; then export "$spec"; fi
  echo "== $spec"; python3 tools/exp_train.py f16 40 0 8192,2000 0 2>&1 | grep "exp_train\] f16"
  if [ "$spec" != "none" ]; then unset "${spec%%=*}"; fi
done
