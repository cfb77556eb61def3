#!/bin/bash
# A/B on one box: bench.py's train leg with the diag library, scatter through the walk alone (MNF_BIN_LEVEL0=16) vs with the bins
export TMPDIR=/tmp
export MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so
mkdir -p gpurun_out/mb
for b in 16 11 16 11; do
  MNF_BIN_LEVEL0=$b timeout 900 python bench.py --workload train --no-cpu-baseline --train-dtypes f16 2> gpurun_out/mb/train_$b.err | tail -1 > gpurun_out/mb/train_$b.json
  python - $b <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/mb/train_{sys.argv[1]}.json").read())
for k in ("train", "train_refyaml"):
    x = (d.get(k) or {}).get("f16")
    if x:
        print("first binned level", sys.argv[1], k, "ms/step %.3f" % x["ms_per_step"], "kept %.0f" % x["rendering_samples_per_step"], "sync %.3f" % x.get("host_synchronous", {}).get("ms_per_step", 0),
              {a: round(b["ms_per_step"], 3) for a, b in x.get("kernels", {}).items()})
PY
done
