#!/usr/bin/env python3
"""Writes tests/golden/mc_cu_v1.npz: CU jobs and the predictions the reference's xeve_mc (src_base/xeve_mc.c:465-610) makes for them.
The reference pictures are not stored: tests regenerate them from the recorded seed (numpy's PCG64 streams are stable).
Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _libs import ptr, ref_mc_cu  # noqa: E402
from _mc_cases import make_jobs, make_refs, refpic_table  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_cu_v1.npz")
R = ref_mc_cu()
d, k = {}, 0
for (seed, w, h, bd, idc, nref, cuw, cuh) in [(501, 64, 48, 10, 1, 2, 8, 8), (502, 64, 48, 10, 1, 2, 16, 16), (503, 96, 64, 10, 1, 2, 32, 32),
                                              (504, 128, 64, 10, 1, 1, 64, 64), (505, 64, 32, 8, 3, 2, 16, 8), (506, 48, 48, 12, 0, 3, 4, 4)]:
    r = np.random.default_rng(seed)
    refs = make_refs(r, w, h, bd, nref, idc)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    jobs = make_jobs(r, 24, w, h, cuw, cuh, nref)
    cw, ch = cuw >> refs["ws"], cuh >> refs["hs"]
    out = [np.zeros((len(jobs), cuw * cuh), np.int16), np.zeros((len(jobs), cw * ch), np.int16), np.zeros((len(jobs), cw * ch), np.int16)]
    for i in range(len(jobs)):
        R.refdrv_mc_cu(ptr(tab), nref, refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), cuw, cuh, bd, bd, idc, ptr(out[0][i]), ptr(out[1][i]), ptr(out[2][i]))
    d["p%d" % k] = np.array([seed, w, h, bd, idc, nref, cuw, cuh])
    d["jobs%d" % k] = jobs.view(np.uint8)
    for c in range(3):
        d["out%d_%d" % (k, c)] = out[c]
    k += 1
d["n"] = np.array(k)
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), k)
