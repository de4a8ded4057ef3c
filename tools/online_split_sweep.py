#!/usr/bin/env python3
"""Which split serves an online launch of a given size best?  (lcm_online.cpp: pick_online_split)
For stored-frame counts n and query rows-per-lane settings (LCM_TUNE_ONLINE_SPLIT: 1, 2, 4 = split mode with that many
query rows per lane; 0 = unsplit, 8 rows per lane), times K queries of 2000 rows against a database of n frames of 2000
rows, two tickets in flight as in bench.py --mode stream — single queries and micro-batches of 8.
    python tools/online_split_sweep.py [--sizes 32,64,...] [--reps 40]
Prints microseconds per launch (wall) and the rate; the records of every setting are compared with split 0's."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="16,32,64,128,256,512,1024,2048,4096,8192")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--desc", type=int, default=2000)
    args = ap.parse_args()
    import __graft_entry__ as entry
    pkg = entry.load_package()
    sizes = [int(x) for x in args.sizes.split(",")]
    fs = pkg.synth.make_frames(max(sizes) + 16, args.desc, seed=pkg.synth.BASE_SEED + 7)
    p = pkg.default_params()
    p.min_gap = 1
    for batch in (1, 8):
        print(f"# {'single queries' if batch == 1 else 'micro-batches of 8'}: us per launch (distances/s) by LCM_TUNE_ONLINE_SPLIT; pairs per launch = n x {batch}")
        with pkg.Matcher(p) as m:
            m.reserve(max(sizes), args.desc)
            stored = 0
            for n in sizes:
                for f in range(stored, n):
                    m.append(int(fs.ids[f]), fs.frame(f))
                stored = n
                qs = [fs.frame(n + i) for i in range(batch)]
                qid = [int(fs.ids[-1]) + 10 + i for i in range(batch)]
                line, ref = f"n = {n:5d}:", None
                for split in (1, 2, 4, 0):
                    m.set_tuning(pkg.capi.TUNE_ONLINE_SPLIT, split)

                    def submit():
                        return m.query_submit(qs[0], qid[0]) if batch == 1 else m.query_submit_batch(qs, qid)

                    def collect(t):
                        return m.query_collect(t)[0] if batch == 1 else m.query_collect_batch(t)[0]

                    got = collect(submit())
                    if ref is None and split == 0:
                        ref = got
                    t_prev = submit()
                    t0 = time.perf_counter()
                    for _ in range(args.reps):
                        t = submit()
                        last = collect(t_prev)
                        t_prev = t
                    collect(t_prev)
                    us = (time.perf_counter() - t0) / args.reps * 1e6
                    line += f"   split {split}: {us:8.1f} us ({n * batch * args.desc * args.desc / (us * 1e-6):.2e})"
                    if split != 0:
                        keep = got
                    else:
                        ref = got
                line += "   " + ("records equal" if np.array_equal(keep, ref) else "RECORDS DIFFER")
                print(line, flush=True)
                m.set_tuning(pkg.capi.TUNE_ONLINE_SPLIT, -1)


if __name__ == "__main__":
    main()
