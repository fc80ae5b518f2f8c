#!/usr/bin/env python3
"""Developer fuzz: xeve_hip_pinter_analyze_cu_jobs against the oracle over random configurations (picture size, bit depth, chroma format, slice type,
reference pictures, QP / lambda extremes, candidate counts, skip_th, sub-pel pattern sizes, candidates far outside the picture)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _inter_cases import fuzz_cases  # noqa: E402
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, oracle_inter, ptr  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402
from test_hip_inter import run_hip  # noqa: E402

O = oracle_inter()
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = total = 0
for refs, org, st, P, jobs, meta in fuzz_cases(n_iter, seed0):
    cu, idc = 1 << meta["lw"], meta["idc"]
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    res, coef, rec, best = run_hip(refs, org, st, P, jobs)
    nc = max(1, (cu >> refs["ws"]) * (cu >> refs["hs"]))
    for i in range(len(jobs)):
        er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
        ec = [np.zeros(cu * cu, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
        ep = [x.copy() for x in ec]
        O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]), ptr(ec[1]),
                               ptr(ec[2]), ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
        ok = res[i:i + 1].tobytes() == er.tobytes() and best[i:i + 1].tobytes() == eb.tobytes()
        for k in range(3 if idc else 1):
            ok = ok and np.array_equal(coef[k][i], ec[k]) and np.array_equal(rec[k][i], ep[k])
        total += 1
        if not ok:
            bad += 1
            if bad <= 5:
                print("MISMATCH", meta, jobs[i], res[i], er[0], flush=True)
print("fuzz done: %d mismatches in %d CUs" % (bad, total))
sys.exit(1 if bad else 0)
