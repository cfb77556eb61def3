"""Per-kernel sums of a rocprofv3 counter_collection CSV (the raw file has one row per dispatch and counter: tens of MB for a scoring pass).
    python tools/sum_pmc.py <counter_collection.csv> > summary.csv        columns: kernel, counter, dispatches, sum"""
import csv
import sys
from collections import defaultdict

tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][:90], r["Counter_Name"])
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "sum"])
for k in sorted(tot, key=lambda k: -tot[k]):
    w.writerow([k[0], k[1], cnt[k], tot[k]])
