#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace of `bench.py --mode stream`: how busy the device was with the library's
kernels, and how much of that time two of its launches (different query slots = different streams) ran side by side.
    stream_trace_summary.py <*_kernel_trace.csv> [steps_to_skip_fraction]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lcm::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
# bench.py --warmup 1 --steps 1 runs two identical passes: the timed one is the second half of the launches
ev = ev[len(ev) // 2:]
span = max(e[1] for e in ev) - ev[0][0]
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
busy = over = 0
depth, last = 0, pts[0][0]
for t, d in pts:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
score = [e for e in ev if "k_score_rowlane" in e[2]]
fold = [e for e in ev if "k_finalize_pairs" in e[2]]
queues = sorted({e[3] for e in ev})
print(f"library kernels in the timed pass: {len(score)} score launches + {len(fold)} k_finalize_pairs on {len(queues)} hardware queues")
print(f"span {span / 1e6:.1f} ms; at least one library kernel running {busy / 1e6:.1f} ms ({100.0 * busy / span:.1f} %); "
      f"two or more side by side {over / 1e6:.1f} ms ({100.0 * over / span:.1f} % of the span)")
print(f"summed kernel durations {sum(e - s for s, e, _, _ in ev) / 1e6:.1f} ms = {sum(e - s for s, e, _, _ in ev) / span:.2f} x the span")
