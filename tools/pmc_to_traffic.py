#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 0 --cpu-seconds 0` into profiles/hbm_traffic.json, which bench.py reports as
roofline.traffic.  Usage: pmc_to_traffic.py <fetch_csv> <write_csv> <out_json> [frames n_gpus workload kernel_substr]"""
import csv
import json
import sys


KERNEL = sys.argv[7] if len(sys.argv) > 7 else "k_score_rowlane<256, 8, 1, false, true>"      # the headline kernel: argmin, packed route


def kernel_sum(path, counter, kernel_substr=KERNEL):   # the loop-search kernel only
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch_kb, nf = kernel_sum(sys.argv[1], "FETCH_SIZE")
write_kb, nw = kernel_sum(sys.argv[2], "WRITE_SIZE")
launches = max(nf, 1)
out = {
    "workload": sys.argv[6] if len(sys.argv) > 6 else "cfg2", "frames": int(sys.argv[4]) if len(sys.argv) > 4 else 1000,
    "n_gpus": int(sys.argv[5]) if len(sys.argv) > 5 else 1, "kernel": KERNEL, "launches_profiled": launches,
    "kernel_variant": 1 if "8, 1, false" in KERNEL else 0,      # bench.py reports the figure only for the matching --variant
    "packed": KERNEL.rstrip(">").endswith("true"),              # ... and the matching route
    "FETCH_SIZE_KB_per_launch": fetch_kb / launches, "WRITE_SIZE_KB_per_launch": write_kb / max(nw, 1),
    # FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024.  This kernel's reads are 64-byte scalar loads (s_load_dwordx16) plus a
    # few 16-byte vector loads, not the 16 B/lane wide streaming pattern for which the guide measured the counter at
    # 1/2 of the true bytes; that x2 correction is therefore NOT applied (uncalibrated access width => raw counter).
    "hbm_bytes_per_launch": (fetch_kb / launches + write_kb / max(nw, 1)) * 1024.0,
    "note": "L2-to-fabric bytes (Infinity-Cache hits are counted, per MI355X_MICROARCH.md); the 64 MB database is "
            "Infinity-Cache resident, so true HBM traffic is lower. No x2 FETCH_SIZE correction: reads are 64-B scalar loads.",
}
if out["packed"]:       # the packed route's second kernel (per-pair fold of the 8 KB per-row scratch), reported beside it
    ffk, fn = kernel_sum(sys.argv[1], "FETCH_SIZE", "k_finalize_bulk")
    fwk, fwn = kernel_sum(sys.argv[2], "WRITE_SIZE", "k_finalize_bulk")
    out["fold_kernel_bytes_per_launch"] = (ffk / max(fn, 1) + fwk / max(fwn, 1)) * 1024.0
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
