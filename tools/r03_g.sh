#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03g; rm -rf $out; mkdir -p $out
(time timeout 400 python bench.py --workload train --no-cpu-baseline) > $out/bench_train.json 2> $out/bench_train.err; echo "bench rc $?"
tail -8 $out/bench_train.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_train.json').read().strip().splitlines()[-1])
t=d['train']
for k in ('f16','bf16'):
    x=t[k]; print(k, x['ms_per_step'], x['rendering_samples_per_step'], x['marched_samples_per_step'], x['skipped_steps'], x['host_synchronous'], x['roofline']['frac'])
    print({a:round(b['ms_per_step'],3) for a,b in x.get('kernels',{}).items()}, x.get('timed_kernels_ms_per_step'), x.get('instrumented_step_ms'))
r=d['train_refyaml']; print('refyaml', r['ms_per_step'], r['rendering_samples_per_step'], r['host_synchronous_ms_per_step'], r.get('fixed_cost_share'))
print({a:round(b['ms_per_step'],3) for a,b in r.get('kernels',{}).items()})
PY
