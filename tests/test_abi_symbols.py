"""CPU suite: libxeve_hip.so loads and exports every symbol include/xeve_hip.h declares (no compute, no GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xeve_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    funcs = set(re.findall(r"\b(xevem?_\w+)\s*\(", src))
    tables = set(re.findall(r"extern\s+const\s+\w+\s+(xevem?_(?:tbl|i?trans_map_tbl)_\w+)\s*\[", src))
    return sorted(funcs), sorted(tables)


def test_library_exports_every_declared_symbol():
    from xeve_amd import lib

    L = lib.load()
    funcs, tables = declared_symbols()
    assert len(funcs) >= 20 and len(tables) == 16
    for name in funcs + tables:
        assert C.c_void_p.in_dll(L, name) is not None, name
    # and the Python binding covers exactly the same set
    assert set(lib.FUNCTIONS) == set(funcs)
    assert set(lib.TABLES) == set(tables)


def test_tables_are_fully_populated():
    from xeve_amd import lib

    L = lib.load()
    for name, tbl in L.tables.items():
        for i, f in enumerate(tbl):
            if name in ("xeve_itrans_map_tbl_hip", "xeve_trans_map_tbl_hip"):  # [16][5]: rows DCT-VIII / DST-VII, sizes 4 .. 32 -- the other slots are NULL in the reference's table too
                assert bool(C.cast(f, C.c_void_p).value) == (i < 10 and i % 5 != 0), (name, i)
            else:
                assert C.cast(f, C.c_void_p).value, name


def test_uninitialised_use_fails_loudly():
    """No silent fallback: the batched API refuses to run before xeve_hip_init() (and there is no GPU here)."""
    from xeve_amd import lib

    L = lib.load()
    rc = L.xeve_hip_avg(None, None, None, 0, None)
    assert rc != 0 and "xeve_hip_init" in lib.last_error()


def test_init_without_gpu_raises():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import xeve_amd

    with pytest.raises(xeve_amd.XeveHipError):
        xeve_amd.init(0)


def test_workspace_sizes_of_the_composites_without_a_gpu():
    """host-side arithmetic only: monotone in the job count, 256 for unusable parameters, and every new composite refuses to run
    before xeve_hip_init()"""
    from xeve_amd import lib

    L = lib.load()
    P = lib.InterParams()
    P.rdo.log2_cuw = P.rdo.log2_cuh = 4
    P.rdo.pic_w, P.rdo.pic_h, P.rdo.slice_type, P.rdo.chroma_format_idc, P.rdo.bit_depth = 128, 64, 0, 1, 10
    P.rdo.num_refp[0] = P.rdo.num_refp[1] = 2
    P.max_cand = 3
    a = L.xeve_hip_pinter_analyze_cu_workspace(100, 4, C.byref(P), 416, 352)
    b = L.xeve_hip_pinter_analyze_cu_workspace(200, 4, C.byref(P), 416, 352)
    assert 256 < a < b
    P.rdo.log2_cuw = 2  # 4x4 CUs are not on this path
    assert L.xeve_hip_pinter_analyze_cu_workspace(100, 4, C.byref(P), 416, 352) == 256
    P.rdo.log2_cuw = 4
    assert 256 < L.xeve_hip_analyze_skip_workspace(100, C.byref(P.rdo), 4) < L.xeve_hip_analyze_skip_workspace(300, C.byref(P.rdo), 4)
    assert L.xeve_hip_analyze_skip_workspace(100, C.byref(P.rdo), 5) == 256
    rc = L.xeve_hip_pinter_analyze_cu_jobs(None, 0, 0, None, 0, 0, None, 0, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, None)
    assert rc != 0 and "xeve_hip_init" in lib.last_error()


def test_python_record_layouts_match_the_library():
    """every ctypes structure / numpy dtype of the binding has the size the library was compiled with (xeve_hip_sizeof; no GPU needed)"""
    import numpy as np

    from xeve_amd import lib

    L = lib.load()
    sz = lambda t: C.sizeof(t) if isinstance(t, type) and issubclass(t, C.Structure) else np.dtype(t).itemsize
    want = {0: 8, 2: lib.MeParams, 3: lib.ME_JOB_DTYPE, 4: lib.ME_RESULT_DTYPE, 5: lib.SpelParams, 6: lib.SPEL_JOB_DTYPE, 7: lib.EPZS_JOB_DTYPE, 8: lib.EpzsParams,
            9: lib.SBAC_DTYPE, 10: lib.CuBitsParams, 11: lib.CU_BITS_JOB_DTYPE, 13: lib.DeblockParams, 14: lib.REFPIC_DTYPE, 15: lib.CU_MC_JOB_DTYPE, 16: lib.RdoParams,
            17: lib.RDO_JOB_DTYPE, 18: lib.RDO_RESULT_DTYPE, 19: lib.SKIP_JOB_DTYPE, 20: lib.SKIP_RESULT_DTYPE, 21: lib.InterParams, 22: lib.INTER_JOB_DTYPE,
            23: lib.INTER_RESULT_DTYPE, 24: lib.IntraParams, 25: lib.INTRA_JOB_DTYPE, 26: lib.INTRA_RESULT_DTYPE, 27: lib.TreeParams, 28: lib.CTU_JOB_DTYPE,
            29: lib.CTU_DATA_DTYPE, 30: lib.TreeInter, 31: lib.EcoParams}
    for i, t in want.items():
        assert L.xeve_hip_sizeof(i) == (t if isinstance(t, int) else sz(t)), (i, t)
    assert L.xeve_hip_sizeof(12) == 4 * lib.EST_FULL_INTS and L.xeve_hip_sizeof(32) == -1


def test_tree_walk_workspace_refuses_what_the_walk_cannot_do():
    """xeve_hip_mode_analyze_ctu_workspace is pure host arithmetic (no GPU): 0 = these parameters are outside the walk, and the walk itself refuses them with the same
    test before its first launch; a supported set gives a size that grows with the number of chains"""
    import ctypes as C

    from xeve_amd import lib

    L = lib.load()

    def params(**kw):
        P = lib.TreeParams()
        P.ip.w_scu, P.ip.h_scu, P.ip.slice_type, P.ip.chroma_format_idc, P.ip.bit_depth = 32, 16, 2, 1, 10
        P.ip.qp[0], P.ip.qp[1], P.ip.qp[2] = 44, 43, 42
        P.ip.lambda_[0] = P.ip.lambda_[1] = P.ip.lambda_[2] = 50.0
        P.ip.sqrt_lambda0, P.ip.dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = 7.07, 1.0, 1.0
        P.pic_w, P.pic_h, P.log2_ctu, P.max_cu, P.min_cu, P.min_cuwh, P.slice_qp = 128, 64, 6, 32, 4, 4, 32
        for k, v in kw.items():
            if k.startswith("ip_"):
                setattr(P.ip, k[3:], v)
            else:
                setattr(P, k, v)
        return P

    ws = lambda P, n=1: L.xeve_hip_mode_analyze_ctu_workspace(n, C.byref(P), None, 128, 64)
    assert 0 < ws(params()) < ws(params(), 8) < ws(params(), 64)
    assert L.xeve_hip_mode_analyze_ctu_intra_workspace(4, C.byref(params())) == ws(params(), 4)
    for bad in (dict(log2_ctu=7), dict(log2_ctu=2), dict(pic_w=130), dict(max_cu=48), dict(min_cu=2), dict(max_cu=4, min_cu=8), dict(ip_w_scu=31), dict(ip_bit_depth=7),
                dict(ip_chroma_format_idc=2), dict(ip_tool_iqt=1), dict(slice_qp=200), dict(ip_slice_type=5)):
        assert ws(params(**bad)) == 0, bad
    P = params()
    P.ip.qp[0] = 80  # above 51 + 6 * (bit depth - 8)
    assert ws(P) == 0
    assert ws(params(), 0) == 0
    # a P / B slice needs its inter side
    assert ws(params(ip_slice_type=0)) == 0 and L.xeve_hip_mode_analyze_ctu_intra_workspace(1, C.byref(params(ip_slice_type=1))) == 0
