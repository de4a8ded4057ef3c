#!/usr/bin/env python3
"""What cross_check costs in the bulk search: 400 x 2000 frames, cross_check 0 / 1 / 2, wall clock and HIP-event time of one
lcm_all_vs_all (run from the repo root on a GPU box: python tools/cross_time.py)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
fs = pkg.synth.make_frames(400, 2000, seed=pkg.synth.BASE_SEED + 2)
p = pkg.default_params(); p.min_gap = 30
with pkg.Matcher(p) as m:
    m.reserve(400, 2000)
    for f in range(400): m.append(int(fs.ids[f]), fs.frame(f))
    n, _ = m.all_vs_all_plan()
    d = m.dev_alloc(n * 8)
    for cc in (0, 1, 2):
        m.set_params(cross_check=cc)
        m.all_vs_all(d, n); m.sync()
        t = time.perf_counter(); m.all_vs_all(d, n); m.sync(); dt = time.perf_counter() - t
        li = m.launch_info()
        print(f"cross_check={cc}: {n} pairs, wall {dt*1e3:.1f} ms, kernel events {li.kernel_ms:.1f} ms, distances {li.distances:.3e}, rate {li.distances/dt/1e12:.3f}e12/s (forward+backward counted)")
