#!/bin/bash
# tools/run_train_variants.sh <name>...  train workload per-kernel times with each gpurun_exp/lib_<name>.so (128x2-only builds)
cd "$(dirname "$0")/.."
for name in "$@"; do
  MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 300 python bench.py --workload train --no-cpu-baseline --steps 20 --train-dtypes f16 --detail-file /tmp/detail_$name.json > /dev/null 2>&1
  python - $name <<'PY'
import json, sys
d = json.load(open(f"/tmp/detail_{sys.argv[1]}.json")); t = d["train"]["f16"]; y = d["train_refyaml"]
fmt = lambda k: " ".join("%s %.3f" % (a, b["ms_per_step"]) for a, b in k.items())
print("%s: 8192 rays %.3f ms/step (kept %.2f M) | %s" % (sys.argv[1], t["ms_per_step"], t["rendering_samples_per_step"] / 1e6, fmt(t["kernels"])))
print("%s: 2000 rays %.3f ms/step (sync %.3f) | %s" % (sys.argv[1], y["ms_per_step"], y["host_synchronous_ms_per_step"], fmt(y["kernels"])))
PY
done
