"""Iterator over tests/golden/sbac_v1.npz (reference outputs of the inter-CU CABAC bit counting)."""
import os

import numpy as np

from _libs import CU_BITS_JOB_DTYPE, SBAC_DTYPE, sbac_from_golden
from _sbac_cases import make_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sbac_v1.npz")


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        lw, lh, st, n0, n1, cm, idc = (int(v) for v in g["p%d" % k])
        jobs = np.ascontiguousarray(g["jobs%d" % k]).view(CU_BITS_JOB_DTYPE)
        n_st = np.ascontiguousarray(g["states%d" % k]).nbytes // 172
        init = np.zeros(n_st, SBAC_DTYPE)
        init["ctx"] = 512  # the models the 172-byte records of this file do not carry start at PROB_INIT
        states = sbac_from_golden(g["states%d" % k], init)
        yield (make_params(lw, lh, st, (n0, n1), cm, idc), states, jobs, np.ascontiguousarray(g["coef%d" % k]),
               sbac_from_golden(g["out%d" % k], states[jobs["sbac"]]), g["bits%d" % k])


def est_states():
    """the coder states of the xeve_rdoq_bit_est golden (172-byte records) in today's layout; the models the file does not carry at PROB_INIT (the estimates
    do not read them)"""
    g = np.load(GOLD)
    raw = np.ascontiguousarray(g["est_states"])
    init = np.zeros(raw.nbytes // 172, SBAC_DTYPE)
    init["ctx"] = 512
    return sbac_from_golden(raw, init)
