"""Deblocking / padding: the oracle against the committed reference goldens (runs without the reference)."""
import numpy as np

from _df_cases import PAD, origin
from _df_golden import golden, golden_pad
from _libs import oracle_df, ptr


def test_oracle_deblock_matches_reference_goldens():
    O = oracle_df()
    n = 0
    for c in golden():
        a = [p.copy() for p in c["planes"]]
        ms = c["map_scu"].copy()
        O.xo_deblock_picture(ptr(a[0], origin(c, 0)), ptr(a[1], origin(c, 1)), ptr(a[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms),
                             ptr(c["map_cu_mode"]), ptr(c["refi"]), ptr(c["mv"]), c["p"])
        for k in range(3):
            assert np.array_equal(a[k], c["out"][k]), (n, k)
        n += 1
    assert n == 6


def test_oracle_picbuf_expand_matches_reference_goldens():
    O = oracle_df()
    for a, out, w, h, e, s in golden_pad():
        b = a.copy()
        O.xo_picbuf_expand(ptr(b, PAD * s + PAD), s, w, h, e)
        assert np.array_equal(b, out)
