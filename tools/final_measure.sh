#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root); results land in gpurun_out/final/.
export TMPDIR=/tmp
out=gpurun_out/final; rm -rf $out; mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $out/pytest_gpu.txt
python bench.py 2>/dev/null | tail -1 > $out/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
python bench.py --views 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_views1.json
python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_train.json
python bench.py --workload score256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_score256.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $out/rocprof_train.err
timeout 300 bash tools/pmc.sh fin_fetch FETCH_SIZE > $out/pmc_fetch.txt 2>&1
timeout 300 bash tools/pmc.sh fin_write WRITE_SIZE > $out/pmc_write.txt 2>&1
cat $out/pytest_gpu.txt; cut -c1-200 $out/bench_line.json
