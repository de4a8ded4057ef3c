#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the loop-search kernels from rocprofv3 --pmc passes of
`bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras` (tools/profile_round.sh) -> profiles/hbm_traffic.json, which
bench.py reports as roofline.traffic (per SCORE LAUNCH, like roofline.achieved).

READS come from the size-resolved request counters of the L2's memory-side interface, which exist on gfx950:
    bytes = 32 x TCC_EA0_RDREQ_32B_sum + 64 x TCC_EA0_RDREQ_64B_sum + 128 x TCC_EA0_RDREQ_128B_sum        (pass `rd`)
Calibrated on known byte counts (tools/fetch_calib: 2^30 bytes read once per pattern, profiles/r03_fetch_calib.json): 1.000 x
for a 16 B/lane stream, for 32-byte rows read as two 16-byte loads (the query rows) and for s_load_dwordx16 lines (the
stored rows).  FETCH_SIZE itself tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md: "exactly 1/2 of a wide
coalesced stream") and is kept only as a cross-check.  WRITES: WRITE_SIZE = 32 / 64-byte write requests x their size.
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
    path = os.path.join(prof, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def read_bytes(kernel_substr):
    """(exact bytes per launch from the size-resolved counters, launches) or (None, 0) if that pass is absent"""
    n32, n = kernel_sum("rd", "TCC_EA0_RDREQ_32B_sum", kernel_substr)
    if n32 is None or n == 0:
        return None, 0
    n64, _ = kernel_sum("rd", "TCC_EA0_RDREQ_64B_sum", kernel_substr)
    n128, _ = kernel_sum("rd", "TCC_EA0_RDREQ_128B_sum", kernel_substr)
    return (32.0 * n32 + 64.0 * n64 + 128.0 * n128) / n, n


argmin = "8, 1, false" in KERNEL
fold = "k_finalize_bulk(" if argmin else "k_finalize_bulk_u16("
fetch_kb, nf = kernel_sum("fetch", "FETCH_SIZE", KERNEL)
write_kb, nw = kernel_sum("write", "WRITE_SIZE", KERNEL)
ffk, fn = kernel_sum("fetch", "FETCH_SIZE", fold)
fwk, fwn = kernel_sum("write", "WRITE_SIZE", fold)
rd, nrd = read_bytes(KERNEL)
frd, _ = read_bytes(fold)
wr = write_kb / max(nw, 1) * 1024.0
fwr = fwk / max(fwn, 1) * 1024.0
fetch_size = fetch_kb / max(nf, 1) * 1024.0
exact = rd is not None
reads = rd if exact else fetch_size
fold_reads = frd if (exact and frd is not None) else ffk / max(fn, 1) * 1024.0
per_launch = reads + wr
out = {
    "workload": workload, "frames": frames, "n_gpus": n_gpus, "kernel": KERNEL, "launches_profiled": max(nrd, nf),
    "launches_per_step": per_step,
    "kernel_variant": 1 if argmin else 0,      # bench.py reports the figure only for the matching --variant
    "packed": KERNEL.rstrip(">").endswith("true"),              # ... and the matching route
    "read_bytes_per_launch": reads,
    "read_bytes_source": ("32 x TCC_EA0_RDREQ_32B_sum + 64 x TCC_EA0_RDREQ_64B_sum + 128 x TCC_EA0_RDREQ_128B_sum (exact: calibrated 1.000 x "
                          "on known byte counts, profiles/r03_fetch_calib.json)") if exact else "FETCH_SIZE (tallies 128-byte requests as 64 bytes)",
    "FETCH_SIZE_bytes_per_launch": fetch_size,
    "write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": per_launch,
    "hbm_bytes_per_step": per_launch * per_step,
    "fold_kernel": fold.rstrip("("),
    "fold_kernel_bytes_per_launch": fold_reads + fwr,
    "step_bytes_score_plus_fold": (per_launch + fold_reads + fwr) * per_step,
    "algorithmic_bytes_per_step": algo_step,
    "traffic_over_algorithmic": (per_launch + fold_reads + fwr) * per_step / algo_step,
    "note": "L2-to-fabric bytes (requests destined for DRAM; Infinity-Cache hits are counted, per MI355X_MICROARCH.md).  Below the "
            "algorithmic bytes because the work items go out slot-run major: the workgroups in flight stream the same stored frames, "
            "so each XCD's L2 fetches a stored frame once per run of slots instead of once per query column; most of what is left is "
            "the query rows (64 KB per work item, 128-byte requests) and the per-row scratch.",
}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
