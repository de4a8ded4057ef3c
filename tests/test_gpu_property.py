"""K11: hypothesis-driven random + adversarial-tie inputs, HIP result == scalar oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


@st.composite
def descriptor_sets(draw):
    nq = draw(st.integers(1, 300))
    nt = draw(st.integers(1, 300))
    seed = draw(st.integers(0, 2**31 - 1))
    alphabet = draw(st.integers(1, 12))           # small alphabets force massive ties
    flips = draw(st.integers(0, 3))
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (alphabet, 32), dtype=np.uint8)
    q = base[rng.integers(0, alphabet, nq)].copy()
    t = base[rng.integers(0, alphabet, nt)].copy()
    for _ in range(flips):                          # a few single-bit perturbations -> near-ties at distance 1, 2
        q[rng.integers(0, nq), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        t[rng.integers(0, nt), rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return q, t


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(descriptor_sets())
def test_match_pair_equals_oracle(matcher, oracle, qt):
    q, t = qt
    idx, d = matcher.match_pair(q, t)
    oi, od = oracle.bf_match(q, t)
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(d.astype(np.int32), od)
    good, md = matcher.match_features(q, t)
    og, omd = oracle.match_features(q, t)
    assert md == omd
    np.testing.assert_array_equal(good["query_idx"], og["query_idx"])
    np.testing.assert_array_equal(good["train_idx"], og["train_idx"])
