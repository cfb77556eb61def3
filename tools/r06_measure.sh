#!/bin/bash
# Round-6 measurement set (run on the GPU box from the repo root); results land in gpurun_out/r06/.
# usage: bash tools/r06_measure.sh [tests] [bench] [benchfull] [prof] [pmc] [pmc_score] [pmc_config2] [pmc_train]   (default: tests bench prof)
export TMPDIR=/tmp
out=gpurun_out/r06; mkdir -p $out
what="${*:-tests bench prof}"
has() { [[ " $what " == *" $1 "* ]]; }
if has tests; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt; fi
if has bench; then timeout 900 python bench.py --detail-file $out/bench_detail.json 2> $out/bench.err > $out/bench_stdout.txt; echo "bench rc $?"; tail -1 $out/bench_stdout.txt > $out/bench_line.json; wc -c $out/bench_line.json; grep "^\[bench" $out/bench.err | tail -25; fi
if has benchfull; then timeout 1500 python bench.py --full --detail-file $out/bench_detail_full.json 2> $out/bench_full.err > $out/bench_full_stdout.txt; echo "bench --full rc $?"; fi
stats() {   # stats <tag> <bench args...>: rocprofv3 kernel stats of one workload
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-timing --detail-file $out/bench_${tag}_under_rocprof.json > /dev/null 2> $out/rocprof_$tag.err
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
  rm -rf $out/prof_$tag
}
if has prof; then
  stats render800_serial --workload render800 --no-views1 --render-jobs 1        # one job in flight: per-launch durations comparable with roofline.avg_launch_ms
  stats train --workload train --steps 20 --warmup 5
  stats score --workload score256 --steps 3
  stats config2 --workload config2 --steps 10
fi
pmc() {   # pmc <tag> <workload args> -- <counters...>
  tag=$1; shift; args=(); while [[ "$1" != "--" ]]; do args+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$tag -- python3 bench.py "${args[@]}" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-views1 --detail-file $out/pmc_$tag.json > /dev/null 2> $out/pmc_$tag.err
  find $out/pmc_$tag -name "*counter_collection.csv" -exec cp {} /tmp/pmc_${tag}_counter_collection.csv \;      # raw rows stay on the box (tens of MB)
  python tools/sum_pmc.py /tmp/pmc_${tag}_counter_collection.csv > $out/pmc_${tag}_per_kernel.csv
  rm -rf $out/pmc_$tag
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
if has pmc; then
  pmc fetch --workload render800 -- FETCH_SIZE
  pmc write --workload render800 -- WRITE_SIZE
  pmc sq --workload render800 -- $SQ
  python tools/reduce_pmc.py /tmp/pmc_fetch_counter_collection.csv /tmp/pmc_write_counter_collection.csv $out/pmc_fetch.json $out/r06_pmc.json "field_kernel<128, 2, 2, false, 0, 0, false" field_kernel
fi
if has pmc_score; then
  pmc score_fetch --workload score256 -- FETCH_SIZE
  pmc score_write --workload score256 -- WRITE_SIZE
  python tools/reduce_pmc.py /tmp/pmc_score_fetch_counter_collection.csv /tmp/pmc_score_write_counter_collection.csv $out/pmc_score_fetch.json $out/r06_pmc.json "field_kernel<128, 2, 2, false, 0, 0, false" field_kernel_scoring
fi
if has pmc_config2; then
  pmc c2_fetch --workload config2 -- FETCH_SIZE
  pmc c2_write --workload config2 -- WRITE_SIZE
  pmc c2_sq --workload config2 -- $SQ
  python tools/reduce_pmc.py /tmp/pmc_c2_fetch_counter_collection.csv /tmp/pmc_c2_write_counter_collection.csv $out/pmc_c2_fetch.json $out/r06_pmc.json "field_kernel<64, 4, 2, false, 0, 0, false" field_kernel_64x4_config2
fi
if has pmc_train; then
  bash tools/pmc_train.sh 8192 f16 128x2@102344280@640/11 r06_pmc_train > $out/pmc_train.txt 2>&1; tail -22 $out/pmc_train.txt
  bash tools/pmc_train.sh 2000 f16 128x2@102344280@640/11 r06_pmc_train_2000 > $out/pmc_train_2000.txt 2>&1; tail -4 $out/pmc_train_2000.txt
  bash tools/pmc_train.sh 8192 bf16 128x2@102344280@640/11 r06_pmc_train_bf16 > $out/pmc_train_bf16.txt 2>&1; tail -4 $out/pmc_train_bf16.txt
  bash tools/pmc_train.sh 2000 f16 64x4@102344250@256/9 r06_pmc_train_64x4_2000 > $out/pmc_train_64x4.txt 2>&1; tail -22 $out/pmc_train_64x4.txt
  cp gpurun_out/r06_pmc_train*.json $out/
fi
ls $out
