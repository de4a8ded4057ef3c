#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes that tools/profile_round.sh collects (FETCH_SIZE | WRITE_SIZE | SQ_* +
GRBM_GUI_ACTIVE, separate runs) into one JSON: per kernel of the library, per counter, sum / dispatches / per launch.
    pmc_summary.py <prof_dir (gpurun_out/<tag>_prof)> <out_json>"""
import csv
import json
import os
import sys

prof, out_path = sys.argv[1], sys.argv[2]
kernels = {}
for sub in ("fetch", "write", "sq", "valu"):
    path = os.path.join(prof, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "lcm::" not in name:
            continue
        c = kernels.setdefault(name, {}).setdefault(r["Counter_Name"], {"sum": 0.0, "dispatches": 0})
        c["sum"] += float(r["Counter_Value"])
        c["dispatches"] += 1
for k in kernels.values():
    for c in k.values():
        c["per_launch"] = c["sum"] / max(c["dispatches"], 1)
out = {
    "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras "
               "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* + GRBM_GUI_ACTIVE | SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_* SQ_WAVE_CYCLES)",
    "units": "FETCH_SIZE / WRITE_SIZE in KB (TCC_EA0 requests x 64 B / 1024), others raw counts; per_launch = sum / dispatches",
    "kernels": kernels,
}
json.dump(out, open(out_path, "w"), indent=1)
for name, k in kernels.items():
    print(name)
    for cn, c in sorted(k.items()):
        print(f"   {cn:18s} per launch {c['per_launch']:.6g}  ({c['dispatches']} dispatches)")
