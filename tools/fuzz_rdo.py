#!/usr/bin/env python3
"""Developer fuzz: xeve_hip_residue_rdo_jobs and xeve_hip_analyze_skip_jobs against the oracle over random configurations (bit depth 8 / 10 / 12,
chroma format, slice type, QP / lambda extremes, vectors far outside the picture with the CU on a picture corner, non-square CUs for the RDO)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _libs import RDO_RESULT_DTYPE, SBAC_DTYPE, SKIP_RESULT_DTYPE, oracle_rdo, oracle_skip, ptr  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402
from _rdo_cases import fuzz_cases  # noqa: E402
from test_hip_rdo import run_hip as run_rdo  # noqa: E402
from test_hip_skip import run_hip as run_skip  # noqa: E402

OR, OS = oracle_rdo(), oracle_skip()
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = total = 0
for refs, org, st, p, lw, lh, jobs, sj, ncand, meta in fuzz_cases(n_iter, seed0):
    cuw, cuh, idc = 1 << lw, 1 << lh, meta["idc"]
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    res, coef, best = run_rdo(refs, org, st, p, jobs)
    nc = max(1, (cuw >> refs["ws"]) * (cuh >> refs["hs"]))
    for i in range(len(jobs)):
        er, eb = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
        ec = [np.zeros(cuw * cuh, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
        OR.xo_residue_rdo(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]),
                          ptr(eb))
        ok = res["cost"][i].tobytes() == er["cost"][0].tobytes() and np.array_equal(res["nnz"][i], er["nnz"][0]) and best[i:i + 1].tobytes() == eb.tobytes()
        for k in range(3 if idc else 1):
            ok = ok and np.array_equal(coef[k][i], ec[k])
        total += 1
        if not ok:
            bad += 1
            if bad <= 5:
                print("RDO MISMATCH", meta, jobs[i], res[i], er[0], flush=True)
    if sj is not None:
        sres, spred, sbest = run_skip(refs, org, st, p, sj, ncand)
        for i in range(len(sj)):
            er, eb = np.zeros(1, SKIP_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ep = [np.zeros(cuw * cuh, np.int16), np.zeros(max(nc, 1), np.int16), np.zeros(max(nc, 1), np.int16)]
            OS.xo_analyze_skip(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(sj[i:i + 1]), ptr(er), ptr(ep[0]), ptr(ep[1]), ptr(ep[2]),
                               ptr(eb))
            ok = sres[i:i + 1].tobytes() == er.tobytes() and sbest[i:i + 1].tobytes() == eb.tobytes() and all(np.array_equal(spred[k][i], ep[k]) for k in range(3 if idc else 1))
            total += 1
            if not ok:
                bad += 1
                if bad <= 5:
                    print("SKIP MISMATCH", meta, sj[i], sres[i], er[0], flush=True)
print("fuzz done: %d mismatches in %d cases" % (bad, total))
sys.exit(1 if bad else 0)
