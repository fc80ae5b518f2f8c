#!/usr/bin/env python3
"""Writes tests/golden/rdoq_v1.npz: inputs and outputs of the reference's xeve_rdoq_run_length_cc
(src_base/xeve_tq.c:497-649, via oracle/ref_rdoq_driver.c).  Build container only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _libs import ptr, ref_rdoq  # noqa: E402
from _rdoq_cases import make_coef, make_est  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rdoq_v1.npz")
R = ref_rdoq()
r = np.random.default_rng(777)
d, k = {}, 0
for (lw, lh) in [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (4, 3), (2, 5)]:
    for it in range(6):
        bd, qp = int(r.choice([8, 10, 10])), int(r.integers(12, 50))
        lam = float(r.choice([0.9, 7.3, 61.0, 410.0])) * (1.0 + float(r.random()))
        luma = int(r.integers(0, 2))
        est = make_est(r)
        c = make_coef(r, lw, lh, bd, it % 4)
        o = c.copy()
        nnz = R.refdrv_rdoq(ptr(o), lw, lh, qp, lam, 0, 0 if luma else 1, bd, 0, C.byref(est))
        d["in%d" % k], d["out%d" % k] = c, o
        d["est%d" % k] = np.frombuffer(bytes(est), dtype=np.int32).copy()
        d["p%d" % k] = np.array([lw, lh, qp, luma, bd, nnz], np.int64)
        d["lam%d" % k] = np.array(lam, np.float64)
        k += 1
d["n"] = np.array(k)
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), k)
