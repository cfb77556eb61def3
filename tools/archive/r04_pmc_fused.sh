#!/bin/bash
# tools/r04_pmc_fused.sh [n_samples]: SQ counters of the fused backward kernel (two rocprofv3 --pmc passes over tools/debug_fused.py) -> gpurun_out/r04_pmc_fused.txt
n=${1:-1000000}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $out/pmcf; mkdir -p $out/pmcf
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/pmcf/p1 -- python3 $GRAFT_REPO_ROOT/tools/debug_fused.py $n 2 > $out/pmcf/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmcf/p2 -- python3 $GRAFT_REPO_ROOT/tools/debug_fused.py $n 2 > $out/pmcf/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r04_pmc_fused.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmcf/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        if 'fused_bwd' in k or 'dgrad' in k or 'wgrad' in k:
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {v / cnt[(k, c)]:16.0f} per launch ({cnt[(k, c)]} launches)')
PY
rm -rf gpurun_out/pmcf/p1 gpurun_out/pmcf/p2
cat gpurun_out/r04_pmc_fused.txt
