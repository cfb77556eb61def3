#!/bin/bash
# A/B of a field-kernel change: field parity tests, then the render bench (trained + random-weight legs, roofline pass)
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "field_forward or render_test_matches or batched_views or bf16_field or repeatable or full_resolution" 2>&1 | tail -3
timeout 600 python bench.py --workload render800 --no-cpu-baseline --steps 10 2> gpurun_out/ab/bench.err | tail -1 > gpurun_out/ab/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ab/bench.json').read().strip().splitlines()[-1])
print('trained: %.2f M rays/s, %.3f G samples/s, %.2f ms | serial field kernel %.4f ms/launch frac %.4f | views1 %.2f M | random: %.2f M rays/s %.3f G samples/s' % (
 d['value']/1e6, d['config']['samples_per_s']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['render_views1']['value']/1e6,
 d['render_random_weights']['value']/1e6, d['render_random_weights']['samples_per_s']/1e9))
PY
