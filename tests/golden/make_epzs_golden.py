#!/usr/bin/env python3
"""Writes tests/golden/me_epzs_v1.npz: (cost, mv, pi->mot_bits) the reference's static pinter_me_epzs (src_base/xeve_pinter.c:699-869, via
oracle/ref_me_driver.c) returns for the seeded cases of tests/_epzs_golden.py.  Build container only."""
import ctypes as C
import os
import sys
from ctypes import c_int, c_void_p

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _epzs_golden import GOLD, cases  # noqa: E402
from _libs import ptr, ref_me  # noqa: E402
from _me_cases import PAD  # noqa: E402

R = ref_me()
R.refdrv_me_epzs_x.restype = C.c_uint32
R.refdrv_me_epzs_x.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, C.c_uint32] + [c_int] * 7 + \
                              [c_void_p, c_void_p, c_int, c_int, c_int]
R.refdrv_me_epzs_mot_bits.restype = C.c_int
rows = []
for c in cases():
    lg = c["S"].bit_length() - 1
    mvp, mv = np.array(c["mvp"], np.int16), np.array(c["mv0"], np.int16)
    mn, mx = np.array(c["min_clip"], np.int32), np.array(c["max_clip"], np.int32)
    cost = R.refdrv_me_epzs_x(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], c["x"], c["y"], lg, lg, 10, ptr(mvp),
                              ptr(mv), c["bi"], c["lambda_mv"], 2, c["refi"], c["mot_other"], c["msr"], c["msr"], c["sr"], 0, ptr(mn), ptr(mx), c["hpel_cnt"], c["qpel_cnt"],
                              2 if c["raster"] else 1)
    rows.append([cost, int(mv[0]), int(mv[1]), R.refdrv_me_epzs_mot_bits(), c["S"], c["bi"], c["raster"], c["refi"], c["hpel_cnt"]])
np.savez_compressed(GOLD, res=np.array(rows, np.int64))
print("wrote", GOLD, os.path.getsize(GOLD))
