#!/bin/bash
# root-cause experiment for the MNF_PK miscompute: each variant several times (the fault is sporadic)
for v in "$@"; do
  for rep in 1 2 3 4 5; do
    echo "== $v run $rep"; MNF_LIB_PATH=$PWD/gpurun_exp/lib_$v.so python tools/debug_pk.py 2>/dev/null | grep -E "^bad|in-kernel|lane |lanes:" | head -16
  done
done
