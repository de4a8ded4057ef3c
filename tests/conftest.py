import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The package directory is `slam-loop-closing_amd` (hyphen): load it by path as slam_loop_closing_amd."""
    name = "slam_loop_closing_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "slam-loop-closing_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    name = "lcm_oracle_binding"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.build()
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture(scope="session")
def matcher(pkg):
    """One GPU matcher for the whole session (GPU tests run in one process)."""
    m = pkg.Matcher()
    yield m
    m.close()


def fast_all_vs_all(oracle, fs, params, shard_rank=0, shard_world=1, check_scalar=12, threads=8):
    """All-vs-all scores in (query asc, stored asc) order through the oracle's TUNED path (itself proven equal to the
    scalar oracle in test_oracle_numpy.py / test_golden.py), plus a scalar spot check — keeps the GPU suite short."""
    import numpy as np
    pq, pt, offs = [], [], [0]
    gap = max(int(params.min_gap), 1)
    for c in range(fs.n_frames):
        for i in range(fs.n_frames):
            if shard_world > 1 and i % shard_world != shard_rank:
                continue
            if fs.ids[c] - fs.ids[i] >= gap:
                pq.append(c); pt.append(i)
        offs.append(len(pq))
    scores, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, pq, pt, params, n_threads=threads)
    rng = np.random.default_rng(len(pq))
    for k in rng.choice(len(pq), size=min(check_scalar, len(pq)), replace=False) if pq else []:
        assert scores[k] == oracle.pair_score(fs.frame(pq[k]), fs.frame(pt[k]), params)
    return scores, np.array(offs, np.int64)


def fast_detect_loops(oracle, fs, cur, params, threads=8):
    import numpy as np
    gap = max(int(params.min_gap), 1)
    elig = [i for i in range(fs.n_frames) if fs.ids[cur] - fs.ids[i] >= gap]
    scores, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, [cur] * len(elig), elig, params, n_threads=threads)
    out = []
    for s, i in zip(scores, elig):
        ok, sim = oracle.loop_test(int(s["good_count"]), int(fs.counts[cur]), int(fs.counts[i]), params)
        if ok:
            out.append((int(fs.ids[cur]), int(fs.ids[i]), int(s["good_count"]), sim))
    return out
