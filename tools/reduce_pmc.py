"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) of the default bench to bytes per evaluated
sample of the field kernel.  usage: reduce_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json> <out.json>"""
import csv, json, sys


def sums(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = "field_kernel" if "field_kernel" in r["Kernel_Name"] else ("round_march_kernel" if "round_march" in r["Kernel_Name"] else None)
        if k:
            tot, n = out.get(k, (0.0, 0))
            out[k] = (tot + float(r["Counter_Value"]), n + 1)
    return out


fetch, write = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
line = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
samples = line["samples"]["process_total"]       # every field-kernel launch of the profiled process (warm-up, timed pass, one-view pass)
f_kb, launches = fetch["field_kernel"]
w_kb, _ = write["field_kernel"]
res = {
    "source": "tools/pmc.sh <tag> FETCH_SIZE ; tools/pmc.sh <tag> WRITE_SIZE (rocprofv3 --kernel-trace --pmc, separate passes, "
              f"bench.py --workload render800 --steps {line['steps']} --warmup {line['warmup']} --no-cpu-baseline --no-kernel-timing); reduced by tools/reduce_pmc.py",
    "field_kernel": {
        "launches": launches, "FETCH_SIZE_KB_sum": f_kb, "WRITE_SIZE_KB_sum": w_kb, "samples_evaluated": samples,
        "fetch_bytes_per_sample_raw": f_kb * 1024 / samples, "write_bytes_per_sample": w_kb * 1024 / samples,
        "hbm_bytes_per_sample": (f_kb + w_kb) * 1024 / samples,
        "note": "FETCH_SIZE = TCC_EA0_RDREQ x 64 B (memory-side requests, Infinity-Cache hits included). MI355X_MICROARCH.md: FETCH_SIZE "
                "under-reports wide coalesced 16 B/lane streams by 2x on gfx950 and is uncalibrated for other widths; this kernel's reads "
                "are random 8-byte gathers, so the raw value is reported. WRITE_SIZE is exact for dword stores.",
    },
    "round_march_kernel": {"FETCH_SIZE_KB_sum": fetch["round_march_kernel"][0], "WRITE_SIZE_KB_sum": write["round_march_kernel"][0],
                           "launches": fetch["round_march_kernel"][1]},
}
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res["field_kernel"], indent=1))
