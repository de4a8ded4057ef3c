#!/usr/bin/env python3
"""Measurement sweep of the bulk search on one GPU: route (plain / packed) x stored frames per work item, both
row-per-lane kernels, cfg2 by default.  Prints one line per setting with the library's own HIP-event kernel times.
    python tools/bulk_sweep.py [--frames 1000] [--desc 2000] [--reps 3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--desc", type=int, default=2000)
    ap.add_argument("--gap", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--slots", default="0,1,2,4,8")
    args = ap.parse_args()
    import __graft_entry__ as entry
    pkg = entry.load_package()
    fs = pkg.synth.make_frames(args.frames, args.desc, seed=pkg.synth.BASE_SEED + 2)
    p = pkg.default_params()
    p.min_gap = args.gap
    with pkg.Matcher(p) as m:
        m.reserve(fs.n_frames, args.desc)
        for f in range(fs.n_frames):
            m.append(int(fs.ids[f]), fs.frame(f))
        n, _ = m.all_vs_all_plan()
        d, ds = m.dev_alloc(n * 8), m.dev_alloc(n * 4)
        ref = None
        for packed in (0, 1, 2):
            for slots in [int(x) for x in args.slots.split(",")]:
                m.set_tuning(pkg.capi.TUNE_PACKED, packed)
                m.set_tuning(pkg.capi.TUNE_ITEM_SLOTS, slots)
                t0, t1, fold = [], [], []
                for _ in range(args.reps):
                    m.all_vs_all(d, n)
                    li = m.launch_info(); t0.append(li.kernel_ms); fold.append(li.aux_kernel_ms)
                    m.all_vs_all_argmin(d, n, ds)
                    t1.append(m.launch_info().kernel_ms)
                got = np.zeros(n, pkg.capi.SCORE_DTYPE)
                m.sync(); m.dev_download(d, got)
                if ref is None:
                    ref = got
                same = bool(np.array_equal(ref, got))
                dist = m.launch_info().distances
                print(f"packed={packed} slots/item={slots or 'auto'}: distance-only {min(t0):8.2f} ms ({dist / min(t0) / 1e9:.4f}e12/s)  "
                      f"argmin {min(t1):8.2f} ms ({dist / min(t1) / 1e9:.4f}e12/s)  fold {np.mean(fold):.2f} ms  "
                      f"workgroups {li.workgroups}  records_equal={same}", flush=True)
        m.dev_free(d); m.dev_free(ds)


if __name__ == "__main__":
    main()
