#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras` into profiles/hbm_traffic.json, which bench.py reports as
roofline.traffic — per SCORE LAUNCH, like roofline.achieved (a packed bulk search is several chunk launches per step).
Usage: pmc_to_traffic.py <prof_dir> <out_json> [frames n_gpus workload kernel_substr launches_per_step algorithmic_bytes_per_step]"""
import csv
import json
import os
import sys

prof, out_path = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
n_gpus = int(sys.argv[4]) if len(sys.argv) > 4 else 1
workload = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
KERNEL = sys.argv[6] if len(sys.argv) > 6 else "k_score_rowlane<256, 8, 1, false, true>"      # the headline kernel: argmin, packed route
per_step = int(sys.argv[7]) if len(sys.argv) > 7 else 9
algo_step = float(sys.argv[8]) if len(sys.argv) > 8 else 30205687480.0


def kernel_sum(sub, counter, kernel_substr):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(os.path.join(prof, sub, "p_counter_collection.csv"))):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch_kb, nf = kernel_sum("fetch", "FETCH_SIZE", KERNEL)
write_kb, nw = kernel_sum("write", "WRITE_SIZE", KERNEL)
launches = max(nf, 1)
argmin = "8, 1, false" in KERNEL
fold = "k_finalize_bulk(" if argmin else "k_finalize_bulk_u16("
ffk, fn = kernel_sum("fetch", "FETCH_SIZE", fold)
fwk, fwn = kernel_sum("write", "WRITE_SIZE", fold)
per_launch = (fetch_kb / launches + write_kb / max(nw, 1)) * 1024.0
out = {
    "workload": workload, "frames": frames, "n_gpus": n_gpus, "kernel": KERNEL, "launches_profiled": launches,
    "launches_per_step": per_step,
    "kernel_variant": 1 if argmin else 0,      # bench.py reports the figure only for the matching --variant
    "packed": KERNEL.rstrip(">").endswith("true"),              # ... and the matching route
    "FETCH_SIZE_KB_per_launch": fetch_kb / launches, "WRITE_SIZE_KB_per_launch": write_kb / max(nw, 1),
    # FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024.  This kernel's reads are 64-byte scalar loads (s_load_dwordx16) plus a
    # few 16-byte vector loads, not the 16 B/lane wide streaming pattern for which the guide measured the counter at
    # 1/2 of the true bytes; that x2 correction is therefore NOT applied (uncalibrated access width => raw counter).
    "hbm_bytes_per_launch": per_launch,
    "hbm_bytes_per_step": per_launch * per_step,
    "fold_kernel": fold.rstrip("("),
    "fold_kernel_bytes_per_launch": (ffk / max(fn, 1) + fwk / max(fwn, 1)) * 1024.0,
    "step_bytes_score_plus_fold": per_launch * per_step + (ffk / max(fn, 1) + fwk / max(fwn, 1)) * 1024.0 * per_step,
    "algorithmic_bytes_per_step": algo_step,
    "traffic_over_algorithmic": (per_launch * per_step + (ffk / max(fn, 1) + fwk / max(fwn, 1)) * 1024.0 * per_step) / algo_step,
    "note": "L2-to-fabric bytes (Infinity-Cache hits are counted, per MI355X_MICROARCH.md).  Below the algorithmic bytes because "
            "the work items go out slot-run major: the workgroups in flight stream the same stored frames, so each XCD's L2 "
            "fetches a stored frame once per run of slots instead of once per query column (round 2, column-major: 1.28 x the "
            "algorithmic bytes).  No x2 FETCH_SIZE correction: reads are 64-B scalar loads.",
}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
