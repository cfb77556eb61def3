#!/bin/bash
# tools/run_train_variants.sh <name>...  train workload per-kernel times with each gpurun_exp/lib_<name>.so
cd "$(dirname "$0")/.."
for name in "$@"; do
  MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 300 python bench.py --workload train --no-cpu-baseline --steps 20 2>/dev/null \
   | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['train']; k=t['kernels']
print('$name: %.2f ms/step kept %.2fM marched %.2fM | ' % (t['ms_per_step'], t['rendering_samples_per_step']/1e6, t['marched_samples_per_step']/1e6) + ' '.join('%s %.3f' % (a, b['ms_per_step']) for a,b in k.items()))"
done
