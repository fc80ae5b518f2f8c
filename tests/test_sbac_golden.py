"""CABAC bit counting: the oracle against the committed reference goldens (runs without the reference)."""
import numpy as np

from _libs import SBAC_DTYPE, oracle_sbac, ptr
from _sbac_golden import golden


def test_oracle_cu_bits_matches_reference_goldens():
    O = oracle_sbac()
    n = 0
    for p, states, jobs, coef, out, bits in golden():
        for i in range(len(jobs)):
            a = np.zeros(1, SBAC_DTYPE)
            assert O.xo_cu_bits(ptr(states), ptr(a), p, ptr(jobs[i:i + 1]), ptr(coef)) == bits[i]
            assert a.tobytes() == out[i:i + 1].tobytes()
            n += 1
    assert n == 133


def test_bit_count_is_the_number_of_renormalisation_shifts():
    """xeve_get_bit_number after xeve_sbac_bit_reset counts one bit per shift of the code register: the property the
    device kernel's fast path relies on"""
    O = oracle_sbac()
    r = np.random.default_rng(5)
    s = np.zeros(1, SBAC_DTYPE)
    O.xo_sbac_reset(ptr(s))
    O.xo_sbac_bit_reset(ptr(s))
    shifts = 0
    for k in range(20000):
        before = int(s["range"][0])
        if r.random() < 0.3:
            O.xo_sbac_bin_ep(ptr(s), int(r.integers(0, 2)))
            shifts += 1
        else:
            ci = int(r.integers(0, 68))
            m = int(s["ctx"][0, ci])
            bit = int(r.random() < 0.2)
            O.xo_sbac_bin(ptr(s), ci, bit)
            lps = max(437, ((m >> 1) * before) >> 9)
            rng = before - lps
            if bit != (m & 1) and rng >= lps:
                rng = lps
            while rng < 8192:
                rng <<= 1
                shifts += 1
            assert rng == int(s["range"][0])
        assert O.xo_sbac_bits(ptr(s)) == shifts


def test_oracle_rdoq_bit_est_matches_reference_goldens():
    from _libs import EST_FULL_INTS
    from _sbac_golden import GOLD, est_states

    O = oracle_sbac()
    g = np.load(GOLD)
    assert [O.xo_entropy_bits(i) for i in range(1024)] == g["entropy_bits"].tolist()
    st = est_states()
    for i in range(len(st)):
        a = np.zeros(EST_FULL_INTS, np.int32)
        O.xo_rdoq_bit_est(ptr(st[i:i + 1]), ptr(a))
        assert np.array_equal(a, g["est"][i]), i
