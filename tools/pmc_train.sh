#!/bin/bash
# tools/pmc_train.sh [rays] [f16|bf16] [model] [tag]: HBM traffic of ONE train step from the hardware counters: separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE;
# TCC_EA0_ATOMIC_sum) over the asynchronous train loop of tools/exp_train.py, reduced per kernel and per step -> gpurun_out/<tag>.json (copy to profiles/).
# The stand-in is trained (and cached) by an un-profiled first run, so the profiled processes run train steps only.
#   default: 8192 rays, f16, 128x2@102344280@640/11 (BASELINE config 5) -> r06_pmc_train.json; bench.py looks for profiles/r06_pmc_train[_bf16][_64x4][_<rays>].json
export TMPDIR=/tmp
R=${1:-8192}; DT=${2:-f16}; MODEL=${3:-128x2@102344280@640/11}; TAG=${4:-r06_pmc_train}; STEPS=12
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
python3 tools/exp_train.py $DT 2 0 $R 0 0 0 $MODEL > $out/${TAG}_warm.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  rm -rf /tmp/pmc_tr_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_tr_$c -- python3 $GRAFT_REPO_ROOT/tools/exp_train.py $DT $STEPS 0 $R 0 0 0 $MODEL > $out/${TAG}_$c.txt 2>&1)
  f=$(find /tmp/pmc_tr_$c -name "*counter_collection.csv" | head -1)
  python3 tools/sum_pmc.py $f > /tmp/pmc_tr_$c.csv
done
python3 tools/reduce_pmc_train.py /tmp/pmc_tr_FETCH_SIZE.csv /tmp/pmc_tr_WRITE_SIZE.csv /tmp/pmc_tr_TCC_EA0_ATOMIC_sum.csv $out/${TAG}_FETCH_SIZE.txt $((STEPS + 5)) $R $out/$TAG.json
