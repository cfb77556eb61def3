#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03f; rm -rf $out; mkdir -p $out
python __graft_entry__.py smoke 2>&1 | tail -6 | tee $out/smoke.txt
(time timeout 1200 python bench.py) > $out/bench_line.json 2> $out/bench.err; echo "bench rc $?"
tail -25 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench_line.json').read().strip().splitlines()[-1])
print(json.dumps({k:v for k,v in d.items() if k not in ('config',)},indent=1)[:9000])
PY
