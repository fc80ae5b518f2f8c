"""Main-profile first slice: the oracle's restatements against the UNMODIFIED Main-profile reference library (oracle/_ref/libxevem_ref.so, built
by oracle/Makefile from src_base + src_main as they lie) -- its exported tables, C and SSE variants."""
import ctypes as C

import numpy as np
import pytest

from _libs import oracle
from _main_cases import REF_NAMES, OracleMain, TableMain, ref_main_lib, run_all

pytestmark = pytest.mark.skipif(ref_main_lib() is None, reason="oracle/_ref/libxevem_ref.so not built (no /root/reference here)")


def test_main_coefficient_tables():
    L, O = ref_main_lib(), oracle()
    for name, oname, n in (("xevem_tbl_mc_l_coeff", "xom_mc_l_coeff", 16 * 8), ("xevem_tbl_mc_c_coeff", "xom_mc_c_coeff", 32 * 4)):
        assert bytes((C.c_int16 * n).in_dll(L, name)) == bytes((C.c_int16 * n).in_dll(O, oname)), name
    bl = np.frombuffer((C.c_int16 * 32).in_dll(L, "xeve_tbl_bl_mc_l_coeff"), np.int16).reshape(16, 2)
    assert np.array_equal(bl, np.stack([64 - 4 * np.arange(16), 4 * np.arange(16)], 1))  # the {64 - 4f, 4f} the oracle builds
    for n in (2, 4, 8, 16, 32, 64):  # the Main library transforms with the same matrices as the Baseline one
        m = np.zeros((n, n), np.int8)
        O.xo_dct_matrix(n, m.ctypes.data_as(C.c_void_p))
        assert bytes((C.c_int8 * (n * n)).in_dll(L, "xeve_tbl_tm%d" % n)) == m.tobytes()


def test_ats_matrices_closed_form():
    """all eight matrices of xevem_tbl_tr[DCT8 | DST7][4 .. 32] from the oracle's closed form == the library's table (xevem_tbl.c:421-565)"""
    L, O = ref_main_lib(), oracle()
    T = np.frombuffer((C.c_int8 * (2 * 4 * 1024)).in_dll(L, "xevem_tbl_tr"), np.int8).reshape(2, 4, 1024)
    O.xo_ats_matrix.restype = None
    O.xo_ats_matrix.argtypes = [C.c_int, C.c_int, C.c_void_p]
    for typ in (0, 1):
        for log2n in range(2, 6):
            n = 1 << log2n
            m = np.zeros(n * n, np.int8)
            O.xo_ats_matrix(typ, log2n, m.ctypes.data_as(C.c_void_p))
            assert np.array_equal(m, T[typ, log2n - 2, :n * n]), (typ, n)


@pytest.mark.parametrize("variant", ["c", "sse"])
def test_main_tables(variant):
    m = 4 if variant == "sse" else 1
    a = run_all(OracleMain(), m)
    b = run_all(TableMain(ref_main_lib(), REF_NAMES[variant]), m)
    assert len(a) == len(b) > 450
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), (variant, i)
