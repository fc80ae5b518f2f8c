"""Main profile: the forward ATS passes (DCT-VIII / DST-VII of 4 .. 32 points; xeve_trans_map_tbl, src_main/xevem_tq.c:53-56, 336-680) -- the counterpart of the inverse
passes of tests/test_main_profile.py.  (cpu) the oracle against goldens of the reference's own table; (gpu) xeve_trans_map_tbl_hip against oracle and goldens."""
import ctypes as C
import os

import numpy as np
import pytest

from _libs import oracle
from _main_cases import ats_cases, ptr

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "main_ats_fwd_v1.npz"))["out"]
FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)


def run(fn):
    out = []
    for typ, log2n, line, shift, sl, s2, blk in ats_cases():
        a = np.full(blk.size, -9, np.int16)
        fn(typ, log2n, blk.copy(), a, shift, line, sl, s2)
        out.append(a)
    return np.concatenate(out)


def oracle_fn():
    O = oracle()
    O.xo_trans_ats.restype = None
    O.xo_trans_ats.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return lambda typ, log2n, blk, coef, shift, line, sl, s2: O.xo_trans_ats(typ, log2n, ptr(blk), ptr(coef), shift, line, sl, s2)


def test_oracle_forward_ats_matches_the_reference_goldens():
    got = run(oracle_fn())
    assert got.size == GOLD.size and np.array_equal(got, GOLD)


@pytest.mark.gpu
def test_hip_forward_ats_table_matches_oracle_and_goldens():
    import xeve_amd
    from xeve_amd import lib

    xeve_amd.init(0)
    L = lib.load()
    tbl = (FN * 80).in_dll(L, "xeve_trans_map_tbl_hip")
    assert not any(bool(tbl[t * 5 + n]) for t in range(2, 16) for n in range(5)) and not bool(tbl[0]) and not bool(tbl[5])  # (the reference's table has these NULL as well)
    before = L.xeve_hip_table_calls_main()
    got = run(lambda typ, log2n, blk, coef, shift, line, sl, s2: tbl[typ * 5 + log2n - 1](ptr(blk), ptr(coef), shift, line, sl, s2))
    assert L.xeve_hip_table_calls_main() - before == 256
    assert np.array_equal(got, GOLD) and np.array_equal(got, run(oracle_fn()))
