"""Per-kernel PMC sums of the train loop (tools/sum_pmc.py rows of three separate passes) -> HBM bytes per train step.
    python tools/reduce_pmc_train.py <fetch.csv> <write.csv> <atomic.csv> <exp_train stdout> <steps in the process> <rays> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived metrics: TCC_EA0_RDREQ-based, 32 B / 64 B requests; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide
coalesced 16-B-per-lane streams by 2x on gfx950 — both the raw and the corrected figure are written, the correction applied to the kernels whose reads ARE such
streams (wgrad, dgrad, bins, optimizer, fills; not the gather kernels, whose reads are 8-byte random gathers or 128-byte rows)."""
import csv
import hashlib
import json
import os
import re
import sys

fetch_csv, write_csv, atomic_csv, log_txt, steps, rays, out_json = sys.argv[1:8]
steps = int(steps)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN_SOURCES = ("train.hip", "trainstep.hip", "composite_train.hip", "field.hip", "field_dev.h", "field.h", "common.h", "march.hip", "march_dev.h")


def rows(path):
    return {r["kernel"]: (int(r["dispatches"]), float(r["sum"])) for r in csv.DictReader(open(path))}


def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"mnf::(f16|bf16)::", "", k)
    return k.replace("mnf::", "")[:60]


STREAMING = ("wgrad", "dgrad", "bin_items", "bin_accumulate", "adam", "count_nan", "fill", "fold_replicas", "composite_train", "compact", "wgrad_reduce")
f, w, a = rows(fetch_csv), rows(write_csv), rows(atomic_csv)
kept = None
for line in open(log_txt):
    m = re.search(r"kept (\d+)", line)
    if m:
        kept = int(m.group(1))
kern = {}
tot = {"fetch_raw": 0.0, "fetch_corrected": 0.0, "write": 0.0, "atomic_requests": 0.0}
for k in sorted(set(f) | set(w) | set(a)):
    fk, wk, ak = f.get(k, (0, 0.0)), w.get(k, (0, 0.0)), a.get(k, (0, 0.0))
    per = lambda v: v / steps
    fr = per(fk[1]) * 1024
    corr = 2.0 if any(s in k for s in STREAMING) else 1.0
    e = {"launches_per_step": max(fk[0], wk[0], ak[0]) / steps, "fetch_bytes_per_step_raw": fr, "fetch_correction": corr,
         "fetch_bytes_per_step": fr * corr, "write_bytes_per_step": per(wk[1]) * 1024, "atomic_requests_per_step": per(ak[1])}
    if e["fetch_bytes_per_step_raw"] + e["write_bytes_per_step"] < 1e6 and e["atomic_requests_per_step"] < 1e4:
        continue
    kern[short(k)] = e
    tot["fetch_raw"] += fr; tot["fetch_corrected"] += fr * corr; tot["write"] += e["write_bytes_per_step"]; tot["atomic_requests"] += e["atomic_requests_per_step"]
h = hashlib.md5()
for s in TRAIN_SOURCES:
    h.update(open(os.path.join(REPO, "active-perception-using-neural-radiance-fields_amd", "csrc", s), "rb").read())
out = {"train_sources_md5": h.hexdigest()[:12], "rays_per_step": int(rays), "steps_in_process": steps, "surviving_samples_per_step": kept,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_ATOMIC_sum (three separate processes) of tools/exp_train.py (asynchronous steps, lr 0, "
                 "stand-in of bench.py's train legs); tools/pmc_train.sh; kernels below 1 MB per step omitted",
       "per_step": {"fetch_bytes_raw": tot["fetch_raw"], "fetch_bytes_corrected": tot["fetch_corrected"], "write_bytes": tot["write"],
                    "atomic_requests": tot["atomic_requests"], "atomic_bytes_64B_rmw": tot["atomic_requests"] * 128,
                    "hbm_bytes": tot["fetch_corrected"] + tot["write"],
                    "note": "memory-side atomics are counted by WRITE_SIZE as their payload; their DRAM read-modify-write (64 B read + 64 B write per request) is listed apart"},
       "kernels": kern}
json.dump(out, open(out_json, "w"), indent=1)
print(json.dumps(out["per_step"], indent=1))
for k, e in sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_step"] + kv[1]["write_bytes_per_step"])):
    print(f"{k:62s} fetch {e['fetch_bytes_per_step'] / 1e6:9.1f} MB  write {e['write_bytes_per_step'] / 1e6:9.1f} MB  atomics {e['atomic_requests_per_step'] / 1e6:7.2f} M")
