#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03e; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $out/tests.txt
timeout 600 python tools/exp_score.py 256,32 5 2>&1 | grep exp_score | tee $out/exp_score.txt
timeout 600 python tools/exp_train.py f16 30 0 8192,2000 1,0 2>&1 | grep exp_train | tee $out/exp_train.txt
timeout 600 python bench.py --workload render800 --no-cpu-baseline --steps 10 2> $out/bench.err | tail -1 > $out/bench_render.json; cut -c1-400 $out/bench_render.json
