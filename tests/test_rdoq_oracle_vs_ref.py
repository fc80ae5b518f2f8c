"""Pins the oracle's RDOQ (xo_rdoq), zig-zag scan and err_scale against the reference's xeve_rdoq_run_length_cc
(src_base/xeve_tq.c:497-649), xeve_tbl_scan and xeve_init_err_scale, through oracle/ref_rdoq_driver.c."""
import ctypes as C

import numpy as np
import pytest

from _libs import oracle_rdoq, ptr, ref_rdoq
from _rdoq_cases import make_coef, make_est

pytestmark = pytest.mark.skipif(ref_rdoq() is None, reason="oracle/_ref/libref_rdoq.so not built (needs /root/reference)")


def test_zigzag_and_err_scale():
    O, R = oracle_rdoq(), ref_rdoq()
    for lw in range(1, 7):
        for lh in range(1, 7):
            n = 1 << (lw + lh)
            s = np.zeros(n, np.uint16)
            O.xo_zigzag(lw, lh, ptr(s))
            assert np.array_equal(s, np.ctypeslib.as_array(R.refdrv_scan(lw, lh), shape=(n,))), (lw, lh)
    for bd in (8, 10, 12):
        for iqt in (0, 1):
            for q in range(6):
                for ls in range(1, 7):
                    assert O.xo_err_scale(q, ls, bd, iqt) == R.refdrv_err_scale(q, ls, bd, iqt), (bd, iqt, q, ls)


@pytest.mark.parametrize("lw,lh", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 4), (5, 2), (6, 5)])
def test_rdoq_matches_reference(lw, lh):
    O, R = oracle_rdoq(), ref_rdoq()
    r = np.random.default_rng(1300 + lw * 8 + lh)
    nz_total = 0
    for it in range(40):
        bd = int(r.choice([8, 10, 10, 12]))
        qp = int(r.integers(10, 52))
        lam = float(r.choice([0.57, 4.3, 37.1, 220.5, 1500.25])) * (1.0 + float(r.random()))
        luma = int(r.integers(0, 2))
        est = make_est(r)
        c0 = make_coef(r, lw, lh, bd, it % 4)
        a, b = c0.copy(), c0.copy()
        n_ref = R.refdrv_rdoq(ptr(a), lw, lh, qp, lam, int(r.integers(0, 2)), 0 if luma else 1, bd, 0, C.byref(est))
        n_or = O.xo_rdoq(ptr(b), lw, lh, qp, lam, luma, bd, 0, C.byref(est))
        assert n_ref == n_or and np.array_equal(a, b), (lw, lh, it, qp, bd)
        nz_total += n_or
    assert nz_total > 0
