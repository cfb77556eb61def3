#!/bin/bash
# Experiment: pipelined gather / MLP kernels on two streams (MNF_FIELD_PIPE=chunks,rounds,mlp_waves,gather_blocks) vs the fused kernel
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python bench.py --workload render800 --weights random --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('$*: %.2f ms/step  %.1f Mrays/s  field %.4f ms/round  views1 %.2f ms' % (d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), d.get('render_views1',{}).get('ms_per_view',0)))"; }
run MNF_X=0
run MNF_FIELD_SPLIT=1
for cfg in 2,20,4,768 4,20,4,768 8,20,4,768 4,20,8,768 4,20,4,512 4,20,4,1024 4,20,4,1536 4,20,2,1024 4,20,6,512; do run MNF_FIELD_PIPE=$cfg; done
run MNF_X=0
