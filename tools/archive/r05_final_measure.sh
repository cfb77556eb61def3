#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root); results land in gpurun_out/final/.
# usage: bash tools/final_measure.sh [tests] [bench] [prof] [pmc] [pmc_score] [pmc_train]   (default: all but pmc_train)
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
what="${*:-tests bench prof pmc pmc_score}"
has() { [[ " $what " == *" $1 "* ]]; }
if has tests; then timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt; fi
if has bench; then timeout 900 python bench.py 2> $out/bench.err | tail -1 > $out/bench_line.json; echo "bench rc ${PIPESTATUS[0]}"; cut -c1-300 $out/bench_line.json; fi
stats() {   # stats <tag> <bench args...>: rocprofv3 kernel stats of one workload
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-timing > $out/bench_${tag}_under_rocprof.json 2> $out/rocprof_$tag.err
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
  rm -rf $out/prof_$tag
}
if has prof; then
  stats render800 --workload render800 --no-views1
  stats render800_serial --workload render800 --no-views1 --render-jobs 1        # one job in flight: per-launch durations comparable with roofline.avg_launch_ms
  stats train --workload train --steps 20 --warmup 5
  stats score --workload score256 --steps 3
fi
pmc() {   # pmc <tag> <workload args> -- <counters...>
  tag=$1; shift; args=(); while [[ "$1" != "--" ]]; do args+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$tag -- python3 bench.py "${args[@]}" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-views1 > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  find $out/pmc_$tag -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${tag}_counter_collection.csv \;      # raw rows stay on the box (tens of MB)
  python tools/sum_pmc.py /tmp/pmc_${tag}_counter_collection.csv > $out/pmc_${tag}_per_kernel.csv
  rm -rf $out/pmc_$tag
}
if has pmc; then
  pmc fetch --workload render800 -- FETCH_SIZE
  pmc write --workload render800 -- WRITE_SIZE
  pmc sq --workload render800 -- SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
  python tools/reduce_pmc.py /tmp/pmc_fetch_counter_collection.csv /tmp/pmc_write_counter_collection.csv $out/pmc_fetch.json $out/r05_pmc.json "field_kernel<128, 2, 2, false, 0, 0, false" field_kernel
fi
if has pmc_score; then
  pmc score_fetch --workload score256 -- FETCH_SIZE
  pmc score_write --workload score256 -- WRITE_SIZE
  pmc score_sq --workload score256 -- SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
  python tools/reduce_pmc.py /tmp/pmc_score_fetch_counter_collection.csv /tmp/pmc_score_write_counter_collection.csv $out/pmc_score_fetch.json $out/r05_pmc.json "field_kernel<128, 2, 2, false, 0, 0, false" field_kernel_scoring
fi
if has pmc_train_atomic; then
  pmc train_atomic --workload train --train-dtypes f16 -- TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
fi
ls $out
if has pmc_train; then bash tools/r05_pmc_train.sh 8192 > $out/pmc_train.txt 2>&1; cp gpurun_out/r05_pmc_train_8192.json $out/r05_pmc_train.json; tail -25 $out/pmc_train.txt; fi
