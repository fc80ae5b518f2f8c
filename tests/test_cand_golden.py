"""Candidate derivation: the oracle against the committed reference goldens (runs without the reference)."""
from _cand_golden import CASES, H_SCU, W_SCU, golden
from _libs import oracle_cand, ptr


def test_oracle_inter_candidates_match_reference_goldens():
    O = oracle_cand()
    n = 0
    for c in golden():
        map_scu, tidx, map_mv, c0, c1 = c["maps"]
        for i in range(len(c["jobs"])):
            j = c["jobs"][i:i + 1].copy()
            O.xo_inter_candidates(ptr(map_scu), ptr(tidx), ptr(map_mv), ptr(c0), ptr(c1), W_SCU, H_SCU, c["lw"], c["lw"], c["slice_type"], ptr(j))
            assert j.tobytes() == c["exp"][i:i + 1].tobytes(), (n, i)
        n += 1
    assert n == len(CASES)
