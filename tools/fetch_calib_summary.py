#!/usr/bin/env python3
"""FETCH_SIZE of tools/fetch_calib's kernels (each reads 2^30 bytes once) -> reported / actual per access pattern.
    fetch_calib_summary.py <dir with p_counter_collection.csv> <out_json>"""
import collections, csv, json, os, sys
d, out_path = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
    if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("k_"):
        agg[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"]) * 1024.0
what = {"k_wide16": "16 B per lane, contiguous (the guide's calibrated pattern: 1/2)",
        "k_rowpair": "32-byte row per lane as two 16-byte loads, rows contiguous (query-row loads)",
        "k_scalar64": "s_load_dwordx16, one 64-byte line per wave (stored rows through the scalar unit)",
        "k_gather32": "32-byte row per lane as two 16-byte loads, rows scattered (argmin re-scan)"}
out = {"bytes_read_per_kernel": 2 ** 30, "patterns": {k: {"what": what.get(k, ""), "FETCH_SIZE_bytes": v, "reported_over_actual": v / 2 ** 30} for k, v in agg.items()}}
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out["patterns"].items():
    print(f"{k:12s} FETCH_SIZE {v['FETCH_SIZE_bytes'] / 2**30:.3f} x the bytes read   ({v['what']})")
