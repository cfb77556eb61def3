#!/bin/bash
# Atomic requests of the table-gradient scatter as the hardware counts them (train workload) -> gpurun_out/final/pmc_train_atomic_per_kernel.csv
export TMPDIR=/tmp
out=gpurun_out/final; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum --output-format csv -d $out/pmc_ta -- python3 bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/pmc_train_atomic.json 2> $out/pmc_train_atomic.err
find $out/pmc_ta -name "*counter_collection.csv" -exec cp {} /tmp/pmc_ta.csv \;
python tools/sum_pmc.py /tmp/pmc_ta.csv > $out/pmc_train_atomic_per_kernel.csv
rm -rf $out/pmc_ta
grep -i "hash_bwd\|fold\|Kernel" $out/pmc_train_atomic_per_kernel.csv | head; tail -3 $out/pmc_train_atomic.err
