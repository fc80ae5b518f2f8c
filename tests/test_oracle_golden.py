"""CPU suite: the oracle against the committed golden vectors (made from the compiled reference by
tests/golden/make_golden.py) -- this is what keeps the oracle pinned where oracle/_ref cannot be built."""
import numpy as np

from _golden_check import GOLDEN, run_golden
from _libs import OracleTables, oracle, ptr


def test_oracle_matches_golden_vectors():
    assert run_golden(OracleTables()) > 300


def test_oracle_dct_matrices_match_golden_tables():
    g = np.load(GOLDEN)
    for n in (2, 4, 8, 16, 32, 64):
        m = np.zeros((n, n), np.int8)
        oracle().xo_dct_matrix(n, ptr(m))
        assert np.array_equal(m, g["tm%d" % n])


def test_known_answers():
    """Hand-checkable known answers for the bit-exactness traps listed in SURVEY.md 7.3(4)."""
    O = oracle()
    a = np.full((8, 8), 1023, np.int16)
    b = np.zeros((8, 8), np.int16)
    assert O.xo_sad(8, 8, ptr(a), ptr(b), 8, 8, 10) == (64 * 1023) >> 2  # shift AFTER the sum
    assert O.xo_ssd(8, 8, ptr(a), ptr(b), 8, 8, 10) == 64 * ((1023 * 1023) >> 4)  # shift PER PIXEL
    # SATD of a flat difference: only the DC coefficient (64 * 1023) >> 2, tile normalisation (s + 2) >> 2
    assert O.xo_satd(8, 8, ptr(a), ptr(b), 8, 8, 10) == ((((64 * 1023) >> 2) + 2) >> 2) >> 2
    # forward 64-point transform zeroes the upper half of the spectrum
    x = np.random.default_rng(0).integers(-1023, 1024, size=64 * 64, dtype=np.int16)
    O.xo_trans(ptr(x), 6, 6, 10)
    c = x.reshape(64, 64)
    assert not c[32:, :].any() and not c[:, 32:].any() and c[:32, :32].any()
