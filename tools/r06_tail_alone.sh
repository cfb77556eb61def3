#!/bin/bash
# tools/r06_tail_alone.sh: the three kernels of the train step's tail (wgrad | hash_bwd_walk | bin_items + bin_accumulate) timed together (the product's schedule), and
# each without the others (diagnostic library: MNF_NO_WGRAD=1 drops the weight gradients, MNF_HASH_BWD_LEVELS=0,0 drops the scatter) -> gpurun_out/r06_tail_alone.txt
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
DIAG=$GRAFT_REPO_ROOT/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
python3 tools/exp_train.py f16 2 0 8192 0 > /dev/null 2>&1      # trains and caches the stand-in
run() {   # run <tag> <env...>
  tag=$1; shift
  rm -rf /tmp/ta_$tag
  (cd /tmp && env "$@" MNF_LIB_PATH=$DIAG true) 2>/dev/null
  for kv in "$@"; do export "$kv"; done
  export MNF_LIB_PATH=$DIAG
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ta_$tag -- python3 $GRAFT_REPO_ROOT/tools/exp_train.py f16 30 0 8192 0 > /tmp/ta_$tag.txt 2>&1)
  for kv in "$@"; do unset "${kv%%=*}"; done
  f=$(find /tmp/ta_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag ($*)"; grep "exp_train\] f16" /tmp/ta_$tag.txt
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("wgrad_kernel", "hash_bwd_walk", "bin_items", "bin_accumulate", "dgrad_kernel"):
        if k in n:
            print("   %-16s calls %5s  avg %8.1f us" % (k, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
{
run together MNF_NOTHING=1
run no_wgrad MNF_NO_WGRAD=1
run no_scatter MNF_HASH_BWD_LEVELS=0,0
run no_bins MNF_BIN_LEVEL0=16
} > $out/r06_tail_alone.txt 2>&1
cat $out/r06_tail_alone.txt
