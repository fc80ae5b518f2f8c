#!/usr/bin/env python3
"""Writes tests/golden/intra_v1.npz: the outputs of the reference's own intra analysis (the static pintra_analyze_cu, through oracle/ref_intra_driver.c) on
the seeded cases of tests/_intra_cases.py -- cost (bit pattern), distortion, modes, coded-block counts, coefficients, reconstruction and exit coder state per
job -- plus a checksum of the seeded inputs.  Build container only."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _intra_cases import CASES, GOLDEN, N_JOBS, input_checksum, make_case, run_ref  # noqa: E402

d = {}
for k, case in enumerate(CASES):
    c = make_case(*case)
    res, coef, rec, best = [], [[], [], []], [[], [], []], []
    for i in range(N_JOBS):
        r, co, rc, b = run_ref(c, i)
        res.append(r), best.append(b)
        for j in range(3):
            coef[j].append(co[j]), rec[j].append(rc[j])
    d["res%d" % k], d["best%d" % k] = np.concatenate(res).view(np.uint8), np.concatenate(best).view(np.uint8)
    for j in range(3):
        d["coef%d_%d" % (k, j)], d["rec%d_%d" % (k, j)] = np.stack(coef[j]), np.stack(rec[j])
    d["in_crc%d" % k] = np.array(input_checksum(c), np.uint32)
d["sbac_nctx"] = np.array(72)
np.savez_compressed(GOLDEN, **d)
print(GOLDEN, os.path.getsize(GOLDEN), "bytes")
