"""ctypes access to the CHECKERS used by the tests (never by the product):

  * oracle/libxeve_oracle.so  -- our plain-C restatement (always available; built on demand)
  * oracle/_ref/libxeveb_ref.so -- the unmodified reference compiled in place by oracle/Makefile
    (present in the build container and, prebuilt, on the GPU box; absent -> `ref()` returns None)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libxeve_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libxeveb_ref.so")
REF_APP = os.path.join(ORACLE_DIR, "_ref", "xeveb_app")

c_int, c_void_p, c_i64 = C.c_int, C.c_void_p, C.c_int64
_oracle = None
_ref = None


def ptr(a, elem_off=0):
    """void* to element `elem_off` of a contiguous int16/int32 numpy array."""
    return C.c_void_p(int(a.ctypes.data) + int(elem_off) * int(a.itemsize))


def oracle():
    global _oracle
    if _oracle is None:
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("xeve_oracle.c", "xeve_oracle.h"))
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < src_m:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        L = C.CDLL(ORACLE_SO)
        L.xo_sad.restype = c_int
        L.xo_sad.argtypes = [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int]
        L.xo_ssd.restype = c_i64
        L.xo_ssd.argtypes = L.xo_sad.argtypes
        L.xo_satd.restype = c_int
        L.xo_satd.argtypes = L.xo_sad.argtypes
        L.xo_diff.restype = None
        L.xo_diff.argtypes = [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
        mc = [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]
        L.xo_mc_l.restype = None
        L.xo_mc_l.argtypes = mc
        L.xo_mc_c.restype = None
        L.xo_mc_c.argtypes = mc
        L.xo_avg.restype = None
        L.xo_avg.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]
        L.xo_dct_matrix.restype = None
        L.xo_dct_matrix.argtypes = [c_int, c_void_p]
        for f in (L.xo_tx, L.xo_itx):
            f.restype = None
            f.argtypes = [c_int, c_void_p, c_void_p, c_int, c_int, c_int]
        for f in (L.xo_trans, L.xo_itrans):
            f.restype = None
            f.argtypes = [c_void_p, c_int, c_int, c_int]
        L.xo_quant.restype = c_int
        L.xo_quant.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]
        L.xo_rdoq_zero_test.restype = c_int
        L.xo_rdoq_zero_test.argtypes = L.xo_quant.argtypes
        L.xo_dquant.restype = None
        L.xo_dquant.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        L.xo_recon.restype = None
        L.xo_recon.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]
        L.mc_l_coeff = (C.c_int16 * (16 * 8)).in_dll(L, "xo_mc_l_coeff")
        L.mc_c_coeff = (C.c_int16 * (32 * 4)).in_dll(L, "xo_mc_c_coeff")
        L.quant_scale = (C.c_int * 12).in_dll(L, "xo_quant_scale")
        L.dq_scale = (C.c_int * 6).in_dll(L, "xo_dq_scale")
        _oracle = L
    return _oracle


FN_SAD = C.CFUNCTYPE(c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int)
FN_SSD = C.CFUNCTYPE(c_i64, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int)
FN_DIFF = C.CFUNCTYPE(None, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int)
FN_MC = C.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p)
FN_AVG = C.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int)
FN_TXB = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int, c_int)


class Ref:
    """The reference's own dispatch tables (reference: src_base/xeve_enc.c:722-779 lists them)."""

    def __init__(self, L):
        self.L = L
        self.variants = {}
        for suffix, names in (
            ("c", dict(sad="xeve_tbl_sad_16b", ssd="xeve_tbl_ssd_16b", diff="xeve_tbl_diff_16b", satd="xeve_tbl_satd_16b",
                       mc_l="xeve_tbl_mc_l", mc_c="xeve_tbl_mc_c", txb="xeve_tbl_txb", itxb="xeve_tbl_itxb",
                       avg="xeve_average_16b_no_clip")),
            ("sse", dict(sad="xeve_tbl_sad_16b_sse", ssd="xeve_tbl_ssd_16b_sse", diff="xeve_tbl_diff_16b_sse",
                         satd="xeve_tbl_satd_16b_sse", mc_l="xeve_tbl_mc_l_sse", mc_c="xeve_tbl_mc_c_sse",
                         txb="xeve_tbl_txb", itxb="xeve_tbl_itxb_sse", avg="xeve_average_16b_no_clip_sse")),
            ("avx", dict(sad="xeve_tbl_sad_16b_avx", ssd="xeve_tbl_ssd_16b_sse", diff="xeve_tbl_diff_16b_sse",
                         satd="xeve_tbl_satd_16b_sse", mc_l="xeve_tbl_mc_l_avx", mc_c="xeve_tbl_mc_c_avx",
                         txb="xeve_tbl_txb_avx", itxb="xeve_tbl_itxb_avx", avg="xeve_average_16b_no_clip_sse")),
        ):
            v = type("V", (), {})()
            v.sad = (FN_SAD * 64).in_dll(L, names["sad"])
            v.ssd = (FN_SSD * 64).in_dll(L, names["ssd"])
            v.diff = (FN_DIFF * 64).in_dll(L, names["diff"])
            v.satd = (FN_SAD * 1).in_dll(L, names["satd"])
            v.mc_l = (FN_MC * 4).in_dll(L, names["mc_l"])
            v.mc_c = (FN_MC * 4).in_dll(L, names["mc_c"])
            v.txb = (FN_TXB * 6).in_dll(L, names["txb"])
            v.itxb = (FN_TXB * 6).in_dll(L, names["itxb"])
            v.avg = FN_AVG((names["avg"], L))
            self.variants[suffix] = v
        self.recon = L.xeve_recon_blk
        self.recon.restype = None
        self.recon.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]
        self.mc_l_coeff = (C.c_int16 * (16 * 8)).in_dll(L, "xeve_tbl_mc_l_coeff")
        self.mc_c_coeff = (C.c_int16 * (32 * 4)).in_dll(L, "xeve_tbl_mc_c_coeff")

    def tm(self, n):
        return np.frombuffer((C.c_int8 * (n * n)).in_dll(self.L, "xeve_tbl_tm%d" % n), dtype=np.int8).reshape(n, n).copy()


def ref():
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = Ref(C.CDLL(REF_SO))
    return _ref


def ilog2(v):
    return int(v).bit_length() - 1


class OracleTables:
    """The oracle behind the same Python call interface as xeve_amd.tables.HipTables (checker side only)."""

    def __init__(self):
        self.O = oracle()

    def sad(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.O.xo_sad(w, h, ptr(a, oa), ptr(b, ob), s1, s2, bd)

    def ssd(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.O.xo_ssd(w, h, ptr(a, oa), ptr(b, ob), s1, s2, bd)

    def satd(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.O.xo_satd(w, h, ptr(a, oa), ptr(b, ob), s1, s2, bd)

    def diff(self, w, h, a, oa, b, ob, s1, s2, s_diff, out, bd):
        self.O.xo_diff(w, h, ptr(a, oa), ptr(b, ob), s1, s2, s_diff, ptr(out))

    def mc_l(self, fx, fy, ref, gx, gy, s_ref, s_pred, pred, w, h, bd, coef):
        self.O.xo_mc_l(fx, fy, ptr(ref), gx, gy, s_ref, s_pred, ptr(pred), w, h, bd, ptr(coef))

    def mc_c(self, fx, fy, ref, gx, gy, s_ref, s_pred, pred, w, h, bd, coef):
        self.O.xo_mc_c(fx, fy, ptr(ref), gx, gy, s_ref, s_pred, ptr(pred), w, h, bd, ptr(coef))

    def avg(self, a, b, d, sa, sb, sd, w, h):
        self.O.xo_avg(ptr(a), ptr(b), ptr(d), sa, sb, sd, w, h)

    def tx(self, log2n, src, dst, shift, line, step):
        self.O.xo_tx(log2n, ptr(src), ptr(dst), shift, line, step)

    def itx(self, log2n, src, dst, shift, line, step):
        self.O.xo_itx(log2n, ptr(src), ptr(dst), shift, line, step)

    def trans(self, coef, lw, lh, bd):
        self.O.xo_trans(ptr(coef), lw, lh, bd)

    def itrans(self, coef, lw, lh, bd):
        self.O.xo_itrans(ptr(coef), lw, lh, bd)

    def recon(self, coef, pred, is_coef, cuw, cuh, s_rec, rec, bd):
        self.O.xo_recon(ptr(coef), ptr(pred), is_coef, cuw, cuh, s_rec, ptr(rec), bd)


# ---- integer-pel motion search: oracle structs + the wrapper around the reference's static me_ipel_diamond ----
class MeParams(C.Structure):
    _fields_ = [("lambda_mv", C.c_uint32), ("refi_bits", C.c_int32), ("extra_bits", C.c_int32), ("bi", C.c_int32),
                ("faststep", C.c_int32), ("max_search_range", C.c_int32), ("range_recentre", C.c_int32),
                ("min_clip", C.c_int32 * 2), ("max_clip", C.c_int32 * 2), ("reserved", C.c_int32)]


class MeJob(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("org_off", C.c_int32), ("range", C.c_int16 * 4), ("gmvp", C.c_int16 * 2),
                ("mvi", C.c_int16 * 2), ("beststep_in", C.c_int32)]


class MeResult(C.Structure):
    _fields_ = [("mv", C.c_int16 * 2), ("cost", C.c_uint32), ("beststep", C.c_int32), ("best_mv_bits", C.c_int32)]


REF_ME_SO = os.path.join(ORACLE_DIR, "_ref", "libref_me.so")
_ref_me = None


def ref_me():
    global _ref_me
    if _ref_me is None and os.path.exists(REF_ME_SO):
        L = C.CDLL(REF_ME_SO)
        L.refdrv_me_ipel_diamond.restype = C.c_uint32
        L.refdrv_me_ipel_diamond.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p, c_int, c_int, C.c_uint32, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p, c_int, c_void_p]
        L.refdrv_mv_bits.restype = c_int
        L.refdrv_mv_bits.argtypes = [c_int] * 4
        L.refdrv_refi_bits.restype = c_int
        L.refdrv_refi_bits.argtypes = [c_int] * 2
        _ref_me = L
    return _ref_me


def oracle_me():
    L = oracle()
    L.xo_mv_bits.restype = c_int
    L.xo_mv_bits.argtypes = [c_int, c_int]
    L.xo_me_ipel_diamond.restype = None
    L.xo_me_ipel_diamond.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, C.POINTER(MeJob), c_int, c_int, c_int,
                                     C.POINTER(MeParams), C.POINTER(MeResult)]
    return L


class SpelParams(C.Structure):
    _fields_ = [("lambda_mv", C.c_uint32), ("refi_bits", C.c_int32), ("extra_bits", C.c_int32), ("bi", C.c_int32),
                ("hpel_cnt", C.c_int32), ("qpel_cnt", C.c_int32)]


class SpelJob(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("org_off", C.c_int32), ("gmvp", C.c_int16 * 2), ("mvi", C.c_int16 * 2)]


def oracle_spel():
    L = oracle_me()
    L.xo_me_spel_pattern.restype = None
    L.xo_me_spel_pattern.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, C.POINTER(SpelJob), c_int, c_int, c_int, c_void_p,
                                     C.POINTER(SpelParams), C.POINTER(MeResult)]
    return L


def ref_spel():
    L = ref_me()
    if L is not None and not hasattr(L, "_spel_bound"):
        L.refdrv_me_spel_pattern.restype = C.c_uint32
        L.refdrv_me_spel_pattern.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_int, C.c_uint32, c_int, c_int, c_int, c_int, c_int, c_void_p]
        L._spel_bound = True
    return L


class EpzsParams(C.Structure):
    _fields_ = [("me", MeParams), ("spel", SpelParams)]


def oracle_epzs():
    L = oracle_spel()
    L.xo_me_epzs.restype = C.c_uint32
    L.xo_me_epzs.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                             C.POINTER(EpzsParams)]
    return L


def ref_epzs():
    L = ref_spel()
    if L is not None and not hasattr(L, "_epzs_bound"):
        L.refdrv_me_epzs.restype = C.c_uint32
        L.refdrv_me_epzs.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                     C.c_uint32, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int]
        L._epzs_bound = True
    return L


# ---- RDOQ ----------------------------------------------------------------------------------------------------------
class RdoqEst(C.Structure):
    _fields_ = [("cbf", C.c_int32 * 2), ("run", (C.c_int32 * 2) * 24), ("level", (C.c_int32 * 2) * 24), ("last", (C.c_int32 * 2) * 2)]


REF_RDOQ_SO = os.path.join(ORACLE_DIR, "_ref", "libref_rdoq.so")
_ref_rdoq = None


def ref_rdoq():
    global _ref_rdoq
    if _ref_rdoq is None and os.path.exists(REF_RDOQ_SO):
        L = C.CDLL(REF_RDOQ_SO)
        L.refdrv_rdoq.restype = c_int
        L.refdrv_rdoq.argtypes = [c_void_p, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, c_int, C.POINTER(RdoqEst)]
        L.refdrv_scan.restype = C.POINTER(C.c_uint16)
        L.refdrv_scan.argtypes = [c_int, c_int]
        L.refdrv_err_scale.restype = C.c_longlong
        L.refdrv_err_scale.argtypes = [c_int] * 4
        _ref_rdoq = L
    return _ref_rdoq


def oracle_rdoq():
    L = oracle()
    L.xo_zigzag.restype = None
    L.xo_zigzag.argtypes = [c_int, c_int, c_void_p]
    L.xo_err_scale.restype = c_i64
    L.xo_err_scale.argtypes = [c_int] * 4
    L.xo_rdoq.restype = c_int
    L.xo_rdoq.argtypes = [c_void_p, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, C.POINTER(RdoqEst)]
    return L


# ---- CABAC (SBAC) bit counting ------------------------------------------------------------------------------------
SBAC_NCTX = 72
SBAC_DTYPE = np.dtype([("range", "<u4"), ("code", "<u4"), ("code_bits", "<u4"), ("stacked_ff", "<u4"), ("stacked_zero", "<u4"),
                       ("pending_byte", "<u4"), ("is_pending_byte", "<u4"), ("bitcounter", "<u4"), ("bin_counter", "<u4"),
                       ("ctx", "<u2", (SBAC_NCTX,))])
CU_BITS_JOB_DTYPE = np.dtype([("coef_off", "<i4", (3,)), ("nnz", "<i4", (3,)), ("sbac", "<i4"), ("mvd", "<i2", (2, 2)), ("refi", "i1", (2,)),
                              ("mvp_idx", "u1", (2,)), ("mode", "u1"), ("dir_flag", "u1"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1")])
assert SBAC_DTYPE.itemsize == 180 and CU_BITS_JOB_DTYPE.itemsize == 44
_SBAC_V1 = np.dtype(SBAC_DTYPE.descr[:-1] + [("ctx", "<u2", (68,))])  # the 172-byte state of the goldens made before the intra models were added


def sbac_from_golden(raw, like):
    """coder states stored in a committed golden file -> today's SBAC_DTYPE.  Files made with the 68-model state (no "sbac_nctx" key) hold exit states of
    inter-CU syntax, which never touches the models added since: those are taken from `like`, the entry state of the same job (array of the same length)."""
    raw = np.ascontiguousarray(raw)
    if raw.nbytes % SBAC_DTYPE.itemsize == 0 and raw.nbytes // SBAC_DTYPE.itemsize == len(like):
        return raw.view(SBAC_DTYPE).reshape(-1)
    old = raw.view(_SBAC_V1).reshape(-1)
    assert len(old) == len(like)
    out = np.zeros(len(old), SBAC_DTYPE)
    for f in SBAC_DTYPE.names[:-1]:
        out[f] = old[f]
    out["ctx"][:, :68] = old["ctx"]
    out["ctx"][:, 68:] = like["ctx"][:, 68:]
    return out


class CuBitsParams(C.Structure):
    _fields_ = [("log2_cuw", C.c_int32), ("log2_cuh", C.c_int32), ("slice_type", C.c_int32), ("num_refp", C.c_int32 * 2),
                ("cm_init", C.c_int32), ("chroma_format_idc", C.c_int32)]


REF_SBAC_SO = os.path.join(ORACLE_DIR, "_ref", "libref_sbac.so")
_ref_sbac = None


def ref_sbac():
    global _ref_sbac
    if _ref_sbac is None and os.path.exists(REF_SBAC_SO):
        L = C.CDLL(REF_SBAC_SO)
        L.refdrv_cu_bits.restype = C.c_uint32
        L.refdrv_cu_bits.argtypes = [c_void_p, c_void_p, C.POINTER(CuBitsParams), c_void_p, c_void_p]
        L.refdrv_sbac_bin.restype = None
        L.refdrv_sbac_bin.argtypes = [c_void_p, c_int, C.c_uint32, c_int]
        L.refdrv_run_length_cc.restype = None
        L.refdrv_run_length_cc.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]
        L.refdrv_rdoq_bit_est.restype = None
        L.refdrv_rdoq_bit_est.argtypes = [c_void_p, c_void_p]
        L.refdrv_entropy_bits.restype = None
        L.refdrv_entropy_bits.argtypes = [c_void_p]
        _ref_sbac = L
    return _ref_sbac


def oracle_sbac():
    L = oracle()
    L.xo_sbac_reset.restype = None
    L.xo_sbac_reset.argtypes = [c_void_p]
    L.xo_sbac_bit_reset.restype = None
    L.xo_sbac_bit_reset.argtypes = [c_void_p]
    L.xo_sbac_bits.restype = C.c_uint32
    L.xo_sbac_bits.argtypes = [c_void_p]
    L.xo_sbac_bin.restype = None
    L.xo_sbac_bin.argtypes = [c_void_p, c_int, C.c_uint32]
    L.xo_sbac_bin_ep.restype = None
    L.xo_sbac_bin_ep.argtypes = [c_void_p, C.c_uint32]
    L.xo_eco_run_length_cc.restype = None
    L.xo_eco_run_length_cc.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]
    L.xo_cu_bits.restype = C.c_uint32
    L.xo_cu_bits.argtypes = [c_void_p, c_void_p, C.POINTER(CuBitsParams), c_void_p, c_void_p]
    L.xo_entropy_bits.restype = C.c_int32
    L.xo_entropy_bits.argtypes = [c_int]
    L.xo_rdoq_bit_est.restype = None
    L.xo_rdoq_bit_est.argtypes = [c_void_p, c_void_p]
    L.xo_rdoq_est_select.restype = None
    L.xo_rdoq_est_select.argtypes = [c_void_p, c_int, c_int, C.POINTER(RdoqEst)]
    return L


EST_FULL_INTS = 108  # xo_rdoq_est_full / xeve_hip_rdoq_est_full: cbf_all, cbf_luma, cbf_cb, cbf_cr [2] each, run[24][2], level[24][2], last[2][2]


# ---- deblocking + padding ------------------------------------------------------------------------------------------
class DeblockParams(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("w_scu", C.c_int32), ("h_scu", C.c_int32), ("log2_max_cuwh", C.c_int32),
                ("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32), ("chroma_format_idc", C.c_int32),
                ("qp_u_offset", C.c_int32), ("qp_v_offset", C.c_int32), ("qp_chroma", (C.c_int32 * 100) * 2)]


REF_DF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_df.so")
_ref_df = None


def ref_df():
    global _ref_df
    if _ref_df is None and os.path.exists(REF_DF_SO):
        L = C.CDLL(REF_DF_SO)
        L.refdrv_deblock_picture.restype = c_int
        L.refdrv_deblock_picture.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(DeblockParams)]
        L.refdrv_deblock_picture_tiles.restype = c_int
        L.refdrv_deblock_picture_tiles.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(DeblockParams), c_int, c_int, c_void_p]
        L.refdrv_picbuf_expand.restype = None
        L.refdrv_picbuf_expand.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int] * 9
        _ref_df = L
    return _ref_df


def oracle_df():
    L = oracle()
    L.xo_deblock_picture.restype = None
    L.xo_deblock_picture.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(DeblockParams)]
    L.xo_deblock_picture_tiles.restype = None
    L.xo_deblock_picture_tiles.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(DeblockParams)]
    L.xo_picbuf_expand.restype = None
    L.xo_picbuf_expand.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    return L


# ---- a8: xeve_mc driver ----------------------------------------------------------------------------------------------
REFPIC_DTYPE = np.dtype([("y", "<u8"), ("u", "<u8"), ("v", "<u8"), ("poc", "<i4"), ("pad_", "<i4")])  # xo_refpic / xeve_hip_refpic (pointers)
CU_MC_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 2)), ("refi", "i1", (2,)), ("pad_", "i1", (2,))])
assert REFPIC_DTYPE.itemsize == 32 and CU_MC_JOB_DTYPE.itemsize == 20


def ref_mc_cu():
    L = ref_df()
    if L is not None and not hasattr(L, "_mc_bound"):
        L.refdrv_mc_cu.restype = None
        L.refdrv_mc_cu.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
        L._mc_bound = True
    return L


def oracle_mc_cu():
    L = oracle()
    L.xo_mc_cu.restype = None
    L.xo_mc_cu.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    return L


# ---- pinter_residue_rdo ------------------------------------------------------------------------------------------------
class RdoParams(C.Structure):
    _fields_ = [("log2_cuw", C.c_int32), ("log2_cuh", C.c_int32), ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("slice_type", C.c_int32),
                ("num_refp", C.c_int32 * 2), ("chroma_format_idc", C.c_int32), ("bit_depth", C.c_int32), ("tool_iqt", C.c_int32),
                ("qp", C.c_int32 * 3), ("pad_", C.c_int32), ("lambda_", C.c_double * 3), ("dist_chroma_weight", C.c_double * 2)]


RDO_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 2)), ("mvd", "<i2", (2, 2)), ("refi", "i1", (2,)), ("mvp_idx", "u1", (2,)),
                          ("dir_flag", "u1"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1"), ("pad_", "u1"), ("sbac", "<i4")])
RDO_RESULT_DTYPE = np.dtype([("cost", "<f8"), ("nnz", "<i4", (3,)), ("pad_", "<i4"), ("dist", "<i8", (2, 3))])
assert RDO_JOB_DTYPE.itemsize == 36 and RDO_RESULT_DTYPE.itemsize == 72 and C.sizeof(RdoParams) == 96

REF_RDO_SO = os.path.join(ORACLE_DIR, "_ref", "libref_rdo.so")
_ref_rdo = None


def ref_rdo():
    global _ref_rdo
    if _ref_rdo is None and os.path.exists(REF_RDO_SO):
        L = C.CDLL(REF_RDO_SO)
        L.refdrv_residue_rdo.restype = None
        L.refdrv_residue_rdo.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(RdoParams), c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]
        _ref_rdo = L
    return _ref_rdo


def oracle_rdo():
    L = oracle()
    L.xo_residue_rdo.restype = None
    L.xo_residue_rdo.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(RdoParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]
    return L


# ---- xeve_analyze_skip ----------------------------------------------------------------------------------------------
SKIP_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (2, 4, 2)), ("refi_pred", "i1", (2, 4)), ("ncand", "<i4"), ("sbac", "<i4"),
                           ("ctx_skip", "u1"), ("pad_", "u1", (3,))])
SKIP_RESULT_DTYPE = np.dtype([("cost", "<f8"), ("best_ssd", "<i8"), ("idx0", "<i4"), ("idx1", "<i4"), ("mv", "<i2", (2, 2)), ("refi", "i1", (2,)),
                              ("pad_", "i1", (6,))])
assert SKIP_JOB_DTYPE.itemsize == 60 and SKIP_RESULT_DTYPE.itemsize == 40


def ref_skip():
    L = ref_rdo()
    if L is not None and not hasattr(L, "_skip_bound"):
        L.refdrv_analyze_skip.restype = None
        L.refdrv_analyze_skip.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(RdoParams), c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]
        L._skip_bound = True
    return L


def oracle_skip():
    L = oracle()
    L.xo_analyze_skip.restype = None
    L.xo_analyze_skip.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(RdoParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]
    return L


# ---- xeve_pinter_analyze_cu (the whole inter analysis of a CU) -------------------------------------------------------------------
class InterParams(C.Structure):  # xo_inter_params
    _fields_ = [("rdo", RdoParams), ("me", MeParams), ("spel", SpelParams), ("refi_bits", (C.c_int32 * 8) * 2), ("range_recentre", (C.c_int32 * 8) * 2),
                ("max_cand", C.c_int32), ("poc", C.c_int32), ("col_list_poc0", C.c_int32), ("pad_", C.c_int32), ("skip_th", C.c_double)]


INTER_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (2, 4, 2)), ("mv_col", "<i2", (2,)), ("sbac", "<i4"), ("ctx_skip", "u1"),
                            ("ctx_pred_mode", "u1"), ("pad_", "u1", (2,))])
INTER_RESULT_DTYPE = np.dtype([("cost", "<f8"), ("cost_inter", "<f8", (5,)), ("cu_mode", "<i4"), ("best_idx", "<i4"), ("mv", "<i2", (2, 2)),
                               ("mvd", "<i2", (2, 2)), ("refi", "i1", (2,)), ("mvp_idx", "u1", (2,)), ("nnz", "<i4", (3,)), ("pad_", "<i4", (2,))])
assert INTER_JOB_DTYPE.itemsize == 52 and INTER_RESULT_DTYPE.itemsize == 96 and C.sizeof(InterParams) == 320


def ref_inter():
    L = ref_skip()
    if L is not None and not hasattr(L, "_inter_bound"):
        L.refdrv_pinter_analyze_cu.restype = None
        L.refdrv_pinter_analyze_cu.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(InterParams), c_int] + [c_void_p] * 9
        L._inter_bound = True
    return L


def oracle_inter():
    L = oracle()
    L.xo_pinter_analyze_cu.restype = None
    L.xo_pinter_analyze_cu.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, C.POINTER(InterParams)] + [c_void_p] * 9
    L.xo_check_best_mvp.restype = c_int
    L.xo_check_best_mvp.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, C.c_double, c_void_p]
    return L


def ref_cand():
    L = ref_inter()
    if L is not None and not hasattr(L, "_cand_bound"):
        L.refdrv_inter_candidates.restype = None
        L.refdrv_inter_candidates.argtypes = [c_void_p] * 5 + [c_int] * 5 + [c_void_p]
        L._cand_bound = True
    return L


def oracle_cand():
    L = oracle()
    L.xo_inter_candidates.restype = None
    L.xo_inter_candidates.argtypes = [c_void_p] * 5 + [c_int] * 5 + [c_void_p]
    return L
