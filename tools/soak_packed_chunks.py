#!/usr/bin/env python3
"""Soak of the packed route's chunking: cfg2-sized searches with small scratch sizes (hundreds of chunk launches alternating
between the two streams and the two scratch halves), every pass compared bit for bit — records and index checksums —
with the plain route's result (no scratch, no fold, one stream).  Hunts ordering bugs between the streams.
    python tools/soak_packed_chunks.py [--frames 700] [--passes 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=700)
    ap.add_argument("--desc", type=int, default=2000)
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--mb", default="16,64,256,1024")
    args = ap.parse_args()
    import __graft_entry__ as entry
    pkg = entry.load_package()
    fs = pkg.synth.make_frames(args.frames, args.desc, seed=99, ragged=True, dup_frac=0.1)
    p = pkg.default_params()
    p.min_gap = 5
    bad = 0
    with pkg.Matcher(p) as m:
        m.reserve(fs.n_frames, args.desc)
        for f in range(fs.n_frames):
            m.append(int(fs.ids[f]), fs.frame(f))
        n, _ = m.all_vs_all_plan()
        d, ds = m.dev_alloc(n * 8), m.dev_alloc(n * 4)
        ref, ref_i = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
        m.set_tuning(pkg.capi.TUNE_PACKED, 0)
        m.all_vs_all_argmin(d, n, ds); m.sync(); m.dev_download(d, ref); m.dev_download(ds, ref_i)
        m.set_tuning(pkg.capi.TUNE_PACKED, 1)
        got, got_i = np.zeros(n, pkg.capi.SCORE_DTYPE), np.zeros(n, np.uint32)
        for mb in [int(x) for x in args.mb.split(",")]:
            m.set_tuning(pkg.capi.TUNE_PACKED_SCRATCH_MB, mb)
            for k in range(args.passes):
                got[:] = 0; got_i[:] = 0
                m.dev_upload(d, got); m.dev_upload(ds, got_i)            # poison: every record must be rewritten
                if k % 2:
                    m.all_vs_all(d, n)
                else:
                    m.all_vs_all_argmin(d, n, ds)
                li = m.launch_info()
                m.sync(); m.dev_download(d, got); m.dev_download(ds, got_i)
                ok = np.array_equal(got, ref) and (k % 2 == 1 or np.array_equal(got_i, ref_i))
                bad += 0 if ok else 1
                print(f"scratch {mb:5d} MiB pass {k} ({'distance-only' if k % 2 else 'argmin'}): {li.launches // 2} chunks, {li.kernel_ms:.1f} ms, "
                      f"{'OK' if ok else 'MISMATCH'}", flush=True)
        m.dev_free(d); m.dev_free(ds)
    print("soak done, mismatching passes:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
