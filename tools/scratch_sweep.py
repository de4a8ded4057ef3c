#!/usr/bin/env python3
"""Packed bulk route: per-row scratch per chunk (LCM_TUNE_PACKED_SCRATCH_MB) against time, both row-per-lane kernels.
A search larger than one chunk runs chunk after chunk (score kernel, fold kernel, next chunk); smaller chunks mean a
smaller device footprint and more launch tails.  Prints one line per setting (library's HIP events: whole call, folds).
    python tools/scratch_sweep.py [--frames 1000] [--desc 2000] [--mb 256,512,1024,2048,8192] [--reps 3]"""
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
    ap.add_argument("--mb", default="256,512,1024,2048,8192")
    ap.add_argument("--argmin-only", action="store_true")
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
        m.set_tuning(pkg.capi.TUNE_PACKED, 1)
        for mb in [int(x) for x in args.mb.split(",")]:
            m.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, mb)
            t0, t1, f0, f1 = [], [], [], []
            for _ in range(args.reps):
                if not args.argmin_only:
                    m.all_vs_all(d, n)
                    li = m.launch_info(); t0.append(li.kernel_ms); f0.append(li.aux_kernel_ms)
                m.all_vs_all_argmin(d, n, ds)
                li = m.launch_info(); t1.append(li.kernel_ms); f1.append(li.aux_kernel_ms)
            got = np.zeros(n, pkg.capi.SCORE_DTYPE)
            m.sync(); m.dev_download(d, got)
            if ref is None:
                ref = got
            dist = li.distances
            a = f"distance-only {min(t0):9.2f} ms ({dist / min(t0) / 1e9:.4f}e12/s, folds {np.mean(f0):.2f} ms)  " if t0 else ""
            print(f"scratch {mb:5d} MiB: chunks {li.launches // 2:4d}  {a}"
                  f"argmin {min(t1):9.2f} ms ({dist / min(t1) / 1e9:.4f}e12/s, folds {np.mean(f1):.2f} ms)  "
                  f"records_equal={bool(np.array_equal(ref, got))}", flush=True)
        m.dev_free(d); m.dev_free(ds)


if __name__ == "__main__":
    main()
