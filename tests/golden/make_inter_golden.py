#!/usr/bin/env python3
"""Writes tests/golden/inter_v1.npz: CUs with their neighbour vectors and what the reference's xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047,
via oracle/ref_rdo_driver.c: its own xeve_get_motion / xeve_get_mv_dir, pinter_me_epzs, check_best_mvp, analyze_bi, pinter_residue_rdo) returns for
them -- cost, cu_mode, motion data, core->nnz, coefficients, reconstruction, core->s_next_best.  Fields the reference leaves stale are zeroed
(_inter_cases.mask_unobservable; coefficients of skipped CUs).  Pictures and coder states are regenerated from the recorded seed.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _inter_cases import make_inter_jobs, make_inter_params, make_inter_picture, mask_unobservable  # noqa: E402
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, ptr, ref_inter  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402
from _rdo_cases import states  # noqa: E402
from _inter_golden import CASES, N_JOBS  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inter_v1.npz")
R = ref_inter()
d = {}
for k, (seed, w, h, bd, nref, idc, st_type, lw, skip_th) in enumerate(CASES):
    r = np.random.default_rng(seed)
    refs, org = make_inter_picture(r, w, h, bd, nref, idc, st_type)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 6)
    P = make_inter_params(r, lw, w, h, bd, nref, idc, st_type, refs, skip_th)
    jobs = make_inter_jobs(r, N_JOBS, w, h, 1 << lw, len(st), refs, st_type)
    n0 = 1 << (2 * lw)
    nc = max(1, n0 >> (refs["ws"] + refs["hs"]))
    res, best = np.zeros(len(jobs), INTER_RESULT_DTYPE), np.zeros(len(jobs), SBAC_DTYPE)
    coef = [np.zeros((len(jobs), n0), np.int16), np.zeros((len(jobs), nc), np.int16), np.zeros((len(jobs), nc), np.int16)]
    rec = [np.zeros_like(c) for c in coef]
    for i in range(len(jobs)):
        R.refdrv_pinter_analyze_cu(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab),
                                   refs["s_l"], refs["s_c"], ptr(st), P, refs["gop"], ptr(jobs[i:i + 1]), ptr(res[i:i + 1]), ptr(coef[0][i]), ptr(coef[1][i]),
                                   ptr(coef[2][i]), ptr(rec[0][i]), ptr(rec[1][i]), ptr(rec[2][i]), ptr(best[i:i + 1]))
    res = mask_unobservable(res, st_type)
    for c in range(3):
        coef[c][res["cu_mode"] == 2] = 0
    d["params%d" % k] = np.frombuffer(bytes(P), dtype=np.uint8).copy()
    d["jobs%d" % k], d["res%d" % k], d["best%d" % k] = jobs.view(np.uint8), res.view(np.uint8), best.view(np.uint8)
    for c in range(3):
        d["coef%d_%d" % (k, c)], d["rec%d_%d" % (k, c)] = coef[c], rec[c]
    print(k, np.bincount(res["cu_mode"], minlength=4), [int(x) for x in res["refi"][:, 0][:8]])
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT))
