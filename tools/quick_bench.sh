#!/bin/bash
# tools/quick_bench.sh [extra bench args]: render workload on the random-weight and the trained scene, one summary line each
cd "$(dirname "$0")/.."
for w in random trained; do
timeout 600 python bench.py --workload render800 --weights $w --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('$w: %.2f ms/step  %.1f Mrays/s  field %.4f ms/launch  frac %.3f  views1 %.2f ms' % (d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('frac',0), (640000.0 * 1e3 / d['render_views1_rays_per_s'] if d.get('render_views1_rays_per_s') else 0)))"
done
