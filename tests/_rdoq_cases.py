"""Seeded RDOQ cases: transformed residual blocks + plausible CABAC bit-estimate tables (entropy_bits values,
xeve_mode.c:304-325: -32768 * (log2(p) - 9) for a 9-bit probability state)."""
import math

import numpy as np

from _libs import RdoqEst, oracle, ptr


def entropy_bits(state):  # xeve_init_bits_est, i = state << 1
    p = (512 * ((state << 1) + 0.5)) / 1024
    return int(-32768 * (math.log(p) / math.log(2.0) - 9))


def make_est(r):
    e = RdoqEst()

    def pair():
        st = int(r.integers(1, 511))  # probability state of the context model
        return entropy_bits(st), entropy_bits(512 - st)

    e.cbf[0], e.cbf[1] = pair()
    for i in range(24):
        e.run[i][0], e.run[i][1] = pair()
        e.level[i][0], e.level[i][1] = pair()
    for i in range(2):
        e.last[i][0], e.last[i][1] = pair()
    return e


def make_coef(r, lw, lh, bd, kind):
    """forward-transformed residual of a plausible prediction error (through the oracle's DCT)"""
    n = 1 << (lw + lh)
    if kind == 0:
        resid = r.integers(-(1 << bd) + 1, 1 << bd, size=n)
    elif kind == 1:
        resid = r.integers(-40, 41, size=n)
    elif kind == 2:
        resid = r.integers(-6, 7, size=n)
    else:
        resid = np.zeros(n, np.int64)
        resid[r.integers(0, n, size=max(1, n // 16))] = r.integers(-300, 301, size=max(1, n // 16))
    c = resid.astype(np.int16)
    oracle().xo_trans(ptr(c), lw, lh, bd)
    return c
