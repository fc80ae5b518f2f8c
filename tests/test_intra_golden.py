"""CPU suite: the oracle's intra analysis (xo_pintra_analyze_cu) against the committed outputs of the reference's own pintra_analyze_cu
(tests/golden/intra_v1.npz, made by tests/golden/make_intra_golden.py) -- the pin that travels where oracle/_ref cannot."""
from _intra_cases import N_JOBS, golden, run_oracle, same


def test_oracle_intra_analysis_matches_golden():
    n = 0
    for case, c, exp in golden():
        for i in range(N_JOBS):
            same(run_oracle(c, i), exp[i], c["idc"], (case[0], i))
            n += 1
    assert n == 11 * N_JOBS
