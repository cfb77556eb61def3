#!/bin/bash
# run-to-run variance of bench.py's train leg on one box (+ clocks and host load)
export TMPDIR=/tmp
mkdir -p gpurun_out/var
cat /proc/loadavg
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --workload train --train-dtypes f16 --no-cpu-baseline 2> gpurun_out/var/err_$i.txt | tail -1 > gpurun_out/var/train_$i.json
  python - $i <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/var/train_{sys.argv[1]}.json").read())
x = d["train"]["f16"]
print("run", sys.argv[1], "ms/step %.3f" % x["ms_per_step"], "sync %.3f" % x["host_synchronous"]["ms_per_step"], {a: round(b["ms_per_step"], 3) for a, b in x["kernels"].items()})
PY
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
done
cat /proc/loadavg
