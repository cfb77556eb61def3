#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root); results land in gpurun_out/final/.
export TMPDIR=/tmp
out=gpurun_out/final; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $out/pytest_gpu.txt
timeout 600 python bench.py 2> $out/bench.err | tail -1 > $out/bench_line.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --workload render800 --no-cpu-baseline --no-kernel-timing --no-views1 > $out/bench_under_rocprof.json 2> $out/rocprof.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/bench_train_under_rocprof.json 2> $out/rocprof_train.err
timeout 300 bash tools/pmc.sh fin_fetch FETCH_SIZE > $out/pmc_fetch.txt 2>&1
timeout 300 bash tools/pmc.sh fin_write WRITE_SIZE > $out/pmc_write.txt 2>&1
timeout 300 bash tools/pmc.sh fin_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY > $out/pmc_sq.txt 2>&1
cat $out/pytest_gpu.txt; cut -c1-300 $out/bench_line.json
