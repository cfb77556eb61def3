#!/bin/bash
# tools/run_variants_all.sh <name>...: headline legs (render800, train, score256) with each gpurun_exp/lib_<name>.so (128x2 only builds), two alternations; one line per run
cd "$(dirname "$0")/.."
for rep in 1 2; do
for name in "$@"; do
  for wl in render800 train score256; do
  MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-kernel-timing --train-dtypes f16 --detail-file /tmp/detail_$name.json 2>/dev/null | tail -1 \
   | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$name rep$rep $wl: %.3f ms/step  value %.4g  views1 %.1f M  train %.3f ms  refyaml %.3f ms  score256 %.2f ms  shard8 %.2f ms' % (d['ms_per_step'], d['value'], d.get('render_views1_rays_per_s',0)/1e6, d.get('train_ms',0), d.get('train_refyaml_ms',0), d.get('score256_ms',0), d.get('score256_shard8_ms',0)))"
  done
done; done
