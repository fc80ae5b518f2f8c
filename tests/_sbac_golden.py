"""Iterator over tests/golden/sbac_v1.npz (reference outputs of the inter-CU CABAC bit counting)."""
import os

import numpy as np

from _libs import CU_BITS_JOB_DTYPE, SBAC_DTYPE
from _sbac_cases import make_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sbac_v1.npz")


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        lw, lh, st, n0, n1, cm, idc = (int(v) for v in g["p%d" % k])
        yield (make_params(lw, lh, st, (n0, n1), cm, idc), np.ascontiguousarray(g["states%d" % k]).view(SBAC_DTYPE),
               np.ascontiguousarray(g["jobs%d" % k]).view(CU_BITS_JOB_DTYPE), np.ascontiguousarray(g["coef%d" % k]),
               np.ascontiguousarray(g["out%d" % k]).view(SBAC_DTYPE), g["bits%d" % k])
