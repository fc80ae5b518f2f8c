#!/usr/bin/env python3
"""Writes tests/golden/sbac_v1.npz: inputs and outputs of the reference's CABAC bit counting for an inter CU
(xeve_sbac_bit_reset + xeve_rdo_bit_cnt_cu_inter / _cu_inter_comp / _cu_skip + xeve_get_bit_number, src_base/xeve_mode.c:39-295,
via oracle/ref_sbac_driver.c).  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _libs import SBAC_DTYPE, ptr, ref_sbac  # noqa: E402
from _sbac_cases import clamp_refi, make_jobs, make_params, make_states  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sbac_v1.npz")
R = ref_sbac()
r = np.random.default_rng(4242)
d, k = {}, 0
for (lw, lh, st, nref, cm, idc) in [(2, 2, 0, (2, 2), 0, 1), (3, 3, 0, (2, 2), 0, 1), (4, 4, 0, (2, 2), 0, 1), (5, 5, 1, (1, 0), 0, 1),
                                    (6, 6, 0, (2, 1), 0, 1), (4, 3, 0, (4, 3), 0, 1), (3, 5, 1, (3, 0), 0, 1), (6, 4, 0, (2, 2), 1, 1),
                                    (3, 3, 0, (2, 2), 0, 3), (2, 6, 0, (2, 2), 0, 1)]:
    states = make_states(r, 5)
    p = make_params(lw, lh, st, nref, cm, idc)
    jobs, coef = make_jobs(r, 14 if lw + lh < 11 else 7, lw, lh, len(states), idc, nnz_mode=k & 1)
    clamp_refi(jobs, nref)
    out = np.zeros(len(jobs), SBAC_DTYPE)
    bits = np.zeros(len(jobs), np.uint32)
    for i in range(len(jobs)):
        bits[i] = R.refdrv_cu_bits(ptr(states), ptr(out[i:i + 1]), p, ptr(jobs[i:i + 1]), ptr(coef))
    d["p%d" % k] = np.array([lw, lh, st, nref[0], nref[1], cm, idc], np.int64)
    d["states%d" % k] = states.view(np.uint8)
    d["jobs%d" % k] = jobs.view(np.uint8)
    d["coef%d" % k] = coef
    d["out%d" % k] = out.view(np.uint8)
    d["bits%d" % k] = bits
    k += 1
# xeve_init_bits_est's table and xeve_rdoq_bit_est (static in xeve_mode.c) on 64 coder states
from _libs import EST_FULL_INTS  # noqa: E402

tab = np.zeros(1024, np.int32)
R.refdrv_entropy_bits(ptr(tab))
est_states = make_states(r, 64)
est = np.zeros((64, EST_FULL_INTS), np.int32)
for i in range(64):
    R.refdrv_rdoq_bit_est(ptr(est_states[i:i + 1]), ptr(est[i]))
d["entropy_bits"], d["est_states"], d["est"] = tab, est_states.view(np.uint8), est
d["n"] = np.array(k)
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), k)
