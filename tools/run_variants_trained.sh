#!/bin/bash
# as run_variants.sh but on the trained stand-in (trains once with the first library, cached in /tmp)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for name in "$@"; do
  MNF_LIB_PATH=$PWD/gpurun_exp/lib_$name.so timeout 300 python bench.py --workload render800 --no-cpu-baseline 2>/dev/null \
   | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('$name rep$rep: %.2f ms/step  %.1f Mrays/s  field %.4f ms/launch  frac %.3f  %.1f samples/ray  views1 %.2f ms' % (d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('frac',0), d['config']['samples_per_ray'], (640000.0 * 1e3 / d['render_views1_rays_per_s'] if d.get('render_views1_rays_per_s') else 0)))"
done; done
