"""Pins the oracle's intra analysis (xo_pintra_analyze_cu and its pieces) against the UNMODIFIED reference: the static pintra_analyze_cu compiled in place
by oracle/ref_intra_driver.c (oracle/_ref/libref_intra.so), and the intra CU syntax of the bit counter through oracle/ref_sbac_driver.c."""
import numpy as np
import pytest

from _intra_cases import CASES, N_JOBS, make_case, oracle_intra, ref_intra, run_oracle, run_ref, same
from _libs import ptr

pytestmark = pytest.mark.skipif(ref_intra() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_pintra_analyze_cu(case):
    c = make_case(*case)
    modes = set()
    for i in range(N_JOBS):
        a, b = run_oracle(c, i), run_ref(c, i)
        same(a, b, c["idc"], (case[0], i))
        modes.add(int(a[0]["ipm"][0, 0]))
    assert len(modes) >= 2  # not always the same predictor


def test_neighbours_and_predictors():
    O, R = oracle_intra(), ref_intra()
    r = np.random.default_rng(77)
    for rep in range(200):
        idc = int(r.choice([1, 1, 3]))
        ch = int(r.integers(0, 3))
        lw = int(r.integers(2, 7))
        w_scu, h_scu = 40, 24
        cu = 1 << lw
        cw = cu if ch == 0 or idc == 3 else cu >> 1
        xs, ys = int(r.integers(0, w_scu - cu // 4 + 1)), int(r.integers(0, h_scu - cu // 4 + 1))
        x, y = (xs * 4, ys * 4) if ch == 0 or idc == 3 else (xs * 2, ys * 2)
        s = 200
        plane = r.integers(0, 1024, size=(120, s)).astype(np.int16)
        m = ((r.random(w_scu * h_scu) < 0.8).astype(np.uint32) << 31) | ((r.random(w_scu * h_scu) < 0.5).astype(np.uint32) << 15)
        tidx = (r.random(w_scu * h_scu) < 0.1).astype(np.uint8) if rep % 3 == 0 else np.zeros(w_scu * h_scu, np.uint8)
        cip = int(rep % 2)
        n = 2 * cw
        la, ua, lb, ub = (np.full(n + 8, -3, np.int16) for _ in range(4))
        O.xo_get_nbr(x, y, cw, cw, ptr(plane, y * s + x), s, ptr(m), ptr(tidx), w_scu, h_scu, ch, cip, 10, idc, ptr(la, 4), ptr(ua, 4))
        R.refdrv_get_nbr(x, y, cw, cw, ptr(plane, y * s + x), s, ptr(m), ptr(tidx), w_scu, h_scu, ch, cip, 10, idc, lw, lw, ptr(lb, 4), ptr(ub, 4), n)
        assert np.array_equal(la[3:4 + n], lb[3:4 + n]) and np.array_equal(ua[3:4 + n], ub[3:4 + n]), (rep, ch, lw)
        for ipm in range(5):
            pa, pb = np.zeros(cw * cw, np.int16), np.zeros(cw * cw, np.int16)
            O.xo_ipred(ptr(la, 4), ptr(ua, 4), ptr(pa), ipm, cw, cw)
            R.refdrv_ipred(ptr(lb, 4), ptr(ub, 4), ptr(pb), ipm, cw, cw, int(ch != 0))
            assert np.array_equal(pa, pb), (rep, ipm)
