#!/usr/bin/env python3
"""Wall-clock latency of one matchFeatures call (2000 x 2000 host rows -> DMatch list) through the C ABI."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
rng = np.random.default_rng(0)
with pkg.Matcher() as m:
    for nq, nt in [(2000, 2000), (500, 500), (2000, 20000)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        m.match_pair(q, t)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); m.match_pair(q, t); ts.append(time.perf_counter() - t0)
        info = m.launch_info()
        print(f"match_pair {nq} x {nt}: median wall {np.median(ts)*1e3:.3f} ms, kernel {info.kernel_ms:.3f} ms, workgroups {info.workgroups}")
