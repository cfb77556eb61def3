#!/bin/bash
# first GPU call of round 3: baseline state of the tree + the scoring diagnostics VERDICT r02 asks for
export TMPDIR=/tmp
out=gpurun_out/r03a; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt
MNF_ROUND_LOG=1 timeout 300 python tools/score_roundlog.py 256 2> $out/score_roundlog_256.txt
MNF_ROUND_LOG=1 timeout 300 python tools/score_roundlog.py 32 2> $out/score_roundlog_32.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_score -- python3 bench.py --workload score256 --steps 3 --no-cpu-baseline --no-kernel-timing > $out/bench_score_under_rocprof.json 2> $out/rocprof_score.err
find $out/prof_score -name "*kernel_stats.csv" -exec cp {} $out/score_kernel_stats.csv \;
rm -rf $out/prof_score
cat $out/pytest_gpu.txt; tail -3 $out/score_roundlog_256.txt; head -c 600 $out/bench_score_under_rocprof.json
