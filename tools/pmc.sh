#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (one rocprofv3 --pmc pass of a short bench; kernel-trace only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --workload render800 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-views1 > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'field_kernel' in k or 'composite' in k or 'round_march' in k:
        print(k, {c: (v, cnt[(k, c)]) for c, v in d.items()})
PY
