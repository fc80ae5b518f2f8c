#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Writes tests/golden/main_ats_fwd_v1.npz: the outputs of the reference's own forward ATS passes (xeve_trans_map_tbl of oracle/_ref/libxevem_ref.so,
src_main/xevem_tq.c:53-56) on the seeded blocks of tests/_main_cases.py ats_cases()."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _main_cases import ats_cases, ptr, ref_main_lib  # noqa: E402

FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
tbl = (FN * 80).in_dll(ref_main_lib(), "xeve_trans_map_tbl")
out = []
for typ, log2n, line, shift, sl, s2, blk in ats_cases():
    a, src = np.full(blk.size, -9, np.int16), blk.copy()  # (held in a name: a temporary would be gone before the call reads it)
    tbl[typ * 5 + log2n - 1](ptr(src), ptr(a), shift, line, sl, s2)
    out.append(a)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "main_ats_fwd_v1.npz")
np.savez_compressed(path, out=np.concatenate(out))
print(len(out), "cases,", os.path.getsize(path), "bytes")
