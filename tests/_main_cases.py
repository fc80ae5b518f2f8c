"""Shared cases of the Main-profile first slice: the dispatch-table entries the Main tools add --
xevem_tbl_dmvr_mc_l / _c, xevem_tbl_bl_mc_l (src_main/xevem_mc.c:465-485) and xeve_tbl_tx / xeve_tbl_itx (src_main/xevem_tq.c:702,
xevem_itdq.c:549).  One deterministic case list; every implementation (reference tables, oracle, HIP tables) runs it through `run_all`."""
import ctypes as C
import os

import numpy as np

from _libs import ORACLE_DIR, c_int, c_void_p, oracle, ptr

REFM_SO = os.path.join(ORACLE_DIR, "_ref", "libxevem_ref.so")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "main_v1.npz")
FN_MCM = C.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int)  # XEVEM_MC (xevem_mc.h:45)
FN_TX = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int)  # XEVE_TX / XEVE_ITX (xeve_def.h)
KINDS = ("dmvr_l", "dmvr_c", "bl_l")
PAD = 8


def mc_cases():
    """(kind, w, h, bd, gmv_x, gmv_y, plane, s_ref, origin, s_pred): every fraction class of every kind at the block sizes the callers use, the
    integer part of gmv random (DMVR ignores it, bilinear follows it), 8 / 10 / 12 bit"""
    r = np.random.default_rng(20260927)
    for kind in range(3):
        unit = 32 if kind == 1 else 16
        sizes = ((4, 4), (8, 4), (4, 8), (8, 8), (16, 8), (16, 16), (32, 16), (32, 32), (64, 64), (12, 12), (20, 20)) if kind != 1 else \
            ((2, 2), (4, 2), (4, 4), (8, 4), (8, 8), (16, 16), (32, 32), (6, 6), (10, 10))
        for bd in (10, 8, 12):
            for (w, h) in (sizes if bd == 10 else sizes[1:4]):
                for fx, fy in ((0, 0), (1, 0), (0, 1), (1, 1), (1, 1)):
                    s_ref = w + 2 * PAD + 8 + int(r.integers(0, 4))
                    plane = r.integers(0, 1 << bd, size=(h + 2 * PAD + 8, s_ref)).astype(np.int16)
                    gx = int(r.integers(-3, 4)) * unit + (int(r.integers(1, unit)) if fx else 0)
                    gy = int(r.integers(-3, 4)) * unit + (int(r.integers(1, unit)) if fy else 0)
                    yield kind, w, h, bd, gx, gy, plane, s_ref, (PAD + 3) * s_ref + PAD + 3, w + int(r.integers(0, 3))


def tx_cases():
    """(fwd, log2n, line, shift, src): both passes of the 2-D transforms xeve_trans / xeve_itrans run with tool_iqt (xevem_tq.c:709-718,
    xevem_itdq.c:551-557) at every (N, line), residual-range and full-range amplitudes (the latter exercises the s16 wrap / ITX_CLIP)"""
    r = np.random.default_rng(715)
    bd = 10
    for log2n in range(1, 7):
        n = 1 << log2n
        for log2l in range(1, 7):
            line = 1 << log2l
            for amp in (1023, 32767):
                src = r.integers(-amp, amp + 1, size=n * line, dtype=np.int16)
                yield True, log2n, line, log2n - 1 + bd - 8, src  # first pass: shift1 of the width
                yield True, log2n, line, log2n + 6, src  # second pass: shift2 of the height
                yield False, log2n, line, 7, src  # ITX_SHIFT1
                yield False, log2n, line, 12 - (bd - 8), src  # ITX_SHIFT2


def ats_cases():
    """(type, log2n, line, shift, skip_line, skip_line_2, coef): both passes of xeve_it_MxN_ats_intra (xevem_itdq.c:278-300; shift 7, then 20 - bit depth), every (N, line) of
    4 .. 32, DCT-VIII and DST-VII, without and with skipped lines / inputs, residual-range and full-range amplitudes"""
    r = np.random.default_rng(278)
    for typ in (0, 1):
        for log2n in range(2, 6):
            n = 1 << log2n
            for line in (4, 8, 16, 32):
                for shift in (7, 10):
                    for (sl, s2) in ((0, 0), (line // 2, 0), (0, n // 2), (line // 4, n // 4)):
                        amp = 32767 if (line + n + shift + sl) % 3 == 0 else 2000
                        yield typ, log2n, line, shift, sl, s2, r.integers(-amp, amp + 1, size=n * line, dtype=np.int16)


def sobel_cases():
    """(vertical, w, h, pred, s_pred): the predictions the affine gradient search differentiates (xevem_pinter.c: CU sizes 8 .. 128), 10-bit samples"""
    r = np.random.default_rng(2341)
    for (w, h) in ((8, 8), (16, 8), (8, 16), (16, 16), (32, 16), (32, 32), (64, 32), (64, 64), (128, 64), (128, 128)):
        s_pred = w + int(r.integers(0, 3)) * 4
        pred = r.integers(0, 1024, size=(h, s_pred)).astype(np.int16)
        for vertical in (0, 1):
            yield vertical, w, h, pred, s_pred


def eq_cases():
    """(vertex_num, w, h, residue, d0, d1, eq0): the normal equations of the 4- and the 6-parameter model; the accumulators start non-zero (the function adds)"""
    r = np.random.default_rng(2397)
    for (w, h) in ((8, 8), (16, 16), (32, 16), (64, 64), (128, 128)):
        for vn in (2, 3):
            res = r.integers(-1023, 1024, size=(h, w)).astype(np.int16)
            d0, d1 = (r.integers(-4092, 4093, size=(h, w)).astype(np.int32) for _ in range(2))
            yield vn, w, h, res, d0, d1, r.integers(-1000, 1000, size=(7, 7)).astype(np.int64)


def ang_cases():
    """(group, right, w, h, ipm, bit depth, le, up, ri): every angular mode of every group, with and without the right line, block shapes 4 .. 64; the three neighbour
    lines are indexed -1 .. w + h - 1 (element 0 of the arrays is index -1)"""
    r = np.random.default_rng(456)
    for g, ipms in ((0, range(3, 12)), (1, range(25, 33)), (2, range(13, 24))):
        for right in (0, 1):
            for (w, h) in ((4, 4), (8, 4), (4, 8), (8, 8), (16, 16), (32, 8), (32, 32), (64, 64)):
                for ipm in ipms:
                    bd = 10 if (ipm + w) % 3 else 8
                    yield (g, right, w, h, ipm, bd) + tuple(r.integers(0, 1 << bd, size=w + h + 1).astype(np.int16) for _ in range(3))


def run_all(impl, mult=1):
    """mult: only blocks whose sides are multiples of it (the reference's SSE variants store whole groups of four samples / rows)"""
    out = []
    for kind, w, h, bd, gx, gy, plane, s_ref, org, sp in mc_cases():
        if (w | h) % mult:
            continue
        pred = np.full((h, sp), -7, np.int16)
        unit = 32 if kind == 1 else 16
        impl.mc(kind, (gx & (unit - 1)) != 0, (gy & (unit - 1)) != 0, plane.copy(), org, gx, gy, s_ref, sp, pred, w, h, bd)
        out.append(pred.ravel())
    for fwd, log2n, line, shift, src in tx_cases():
        dst = np.zeros(src.size, np.int16)
        impl.tx(fwd, log2n, src.copy(), dst, shift, line)
        out.append(dst)
    for typ, log2n, line, shift, sl, s2, coef in ats_cases():
        if (line | (1 << log2n)) % mult or (mult > 1 and (sl or s2)):
            continue
        dst = np.full(coef.size, -9, np.int16)
        impl.itrans_ats(typ, log2n, coef.copy(), dst, shift, line, sl, s2)
        out.append(dst)
    for vertical, w, h, pred, s_pred in sobel_cases():
        der = np.full((h, w), -5, np.int32)
        impl.sobel(vertical, pred.copy(), s_pred, der, w, w, h)
        out.append(der.ravel().view(np.int16))
    for g, right, w, h, ipm, bd, le, up, ri in ang_cases():
        dst = np.full(w * h, -3, np.int16)
        impl.ang(g, right, le.copy(), up.copy(), ri.copy(), dst, w, h, ipm, bd)
        out.append(dst)
    for vn, w, h, res, d0, d1, eq0 in eq_cases():
        eq = eq0.copy() if mult == 1 else np.zeros_like(eq0)  # (the reference's SSE variant stores its sums: it needs the zeroed accumulators its caller hands it)
        impl.eq_coef(res.copy(), w, d0.copy(), d1.copy(), w, eq, w, h, vn)
        out.append(eq.ravel().view(np.int16))
    return out


class OracleMain:
    def __init__(self):
        self.O = oracle()
        self.O.xo_mc_main.restype = None
        self.O.xo_mc_main.argtypes = [c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int]

    def mc(self, kind, fx, fy, plane, org, gx, gy, s_ref, sp, pred, w, h, bd):
        self.O.xo_mc_main(kind, int(fx), int(fy), ptr(plane, org), gx, gy, s_ref, sp, ptr(pred), w, h, bd)

    def tx(self, fwd, log2n, src, dst, shift, line):
        (self.O.xo_tx if fwd else self.O.xo_itx)(log2n, ptr(src), ptr(dst), shift, line, 2)

    def itrans_ats(self, typ, log2n, coef, dst, shift, line, sl, s2):
        self.O.xo_itrans_ats.restype = None
        self.O.xo_itrans_ats.argtypes = [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int]
        self.O.xo_itrans_ats(typ, log2n, ptr(coef), ptr(dst), shift, line, sl, s2)

    def ang(self, g, right, le, up, ri, dst, w, h, ipm, bd):
        self.O.xo_ipred_ang.restype = None
        self.O.xo_ipred_ang.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]
        self.O.xo_ipred_ang(g, right, ptr(le, 1), ptr(up, 1), ptr(ri, 1), ptr(dst), w, h, ipm, bd)

    def sobel(self, vertical, pred, s_pred, der, s_der, w, h):
        self.O.xo_sobel.restype = None
        self.O.xo_sobel.argtypes = [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int]
        self.O.xo_sobel(vertical, ptr(pred), s_pred, ptr(der), s_der, w, h)

    def eq_coef(self, res, s_res, d0, d1, s_der, eq, w, h, vn):
        self.O.xo_equal_coeff.restype = None
        self.O.xo_equal_coeff.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int]
        self.O.xo_equal_coeff(ptr(res), ptr(d0), ptr(d1), s_der, ptr(eq), w, h, vn)


FN_ITR = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int, c_int, c_int)  # XEVE_INV_TRANS (xevem_type.h:47)
FN_SOBEL = C.CFUNCTYPE(None, c_void_p, c_int, c_void_p, c_int, c_int, c_int)  # XEVE_AFFINE_H / V_SOBEL_FLT (xevem_mc.h:157-168)
FN_ANG = C.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, C.c_uint16, c_void_p, c_int, c_int, c_int, c_int)  # XEVE_INTRA_PRED_ANG (xevem_ipred.h:104-112)
FN_EQ = C.CFUNCTYPE(None, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int)  # XEVE_AFFINE_EQUAL_COEF (xevem_mc.h:169-176)


class TableMain:
    """any library exporting the five tables under `names` (the reference's C or SIMD variants, or libxeve_hip.so's *_hip tables)"""

    def __init__(self, L, names):
        self.L = L
        self.t = [(FN_MCM * 4).in_dll(L, names[k]) for k in range(3)]
        self.f = (FN_TX * 6).in_dll(L, names[3])
        self.i = (FN_TX * 6).in_dll(L, names[4])
        self.itr = (FN_ITR * 80).in_dll(L, names[5])  # [16][5], rows 0 (DCT-VIII) and 1 (DST-VII) populated
        self.sob = [FN_SOBEL((names[6], L)), FN_SOBEL((names[7], L))]
        self.eq = FN_EQ((names[8], L))
        self.angt = (FN_ANG * 6).in_dll(L, names[9])  # [3][2]

    def mc(self, kind, fx, fy, plane, org, gx, gy, s_ref, sp, pred, w, h, bd):
        self.t[kind][int(fx) * 2 + int(fy)](ptr(plane, org), gx, gy, s_ref, sp, ptr(pred), w, h, bd)

    def tx(self, fwd, log2n, src, dst, shift, line):
        (self.f if fwd else self.i)[log2n - 1](ptr(src), ptr(dst), shift, line)

    def itrans_ats(self, typ, log2n, coef, dst, shift, line, sl, s2):
        self.itr[typ * 5 + log2n - 1](ptr(coef), ptr(dst), shift, line, sl, s2)

    def ang(self, g, right, le, up, ri, dst, w, h, ipm, bd):
        self.angt[g * 2 + right](ptr(le, 1), ptr(up, 1), ptr(ri, 1), 0, ptr(dst), w, h, ipm, bd)

    def sobel(self, vertical, pred, s_pred, der, s_der, w, h):
        self.sob[vertical](ptr(pred), s_pred, ptr(der), s_der, w, h)

    def eq_coef(self, res, s_res, d0, d1, s_der, eq, w, h, vn):
        dd = (c_void_p * 2)(d0.ctypes.data, d1.ctypes.data)  # int **derivate
        self.eq(ptr(res), s_res, dd, s_der, ptr(eq), w, h, vn)


REF_NAMES = {
    "c": ("xevem_tbl_dmvr_mc_l", "xevem_tbl_dmvr_mc_c", "xevem_tbl_bl_mc_l", "xeve_tbl_tx", "xeve_tbl_itx", "xeve_itrans_map_tbl",
          "xevem_scaled_horizontal_sobel_filter", "xevem_scaled_vertical_sobel_filter", "xevem_equal_coeff_computer", "xeve_tbl_intra_pred_ang"),
    "sse": ("xeve_tbl_dmvr_mc_l_sse", "xeve_tbl_dmvr_mc_c_sse", "xeve_tbl_bl_mc_l_sse", "xeve_tbl_tx", "xeve_tbl_itx", "xeve_itrans_map_tbl_sse",
            "xevem_scaled_horizontal_sobel_filter_sse", "xevem_scaled_vertical_sobel_filter_sse", "xevem_equal_coeff_computer_sse", "xeve_tbl_intra_pred_ang"),
}
HIP_NAMES = ("xevem_tbl_dmvr_mc_l_hip", "xevem_tbl_dmvr_mc_c_hip", "xevem_tbl_bl_mc_l_hip", "xeve_tbl_tx_hip", "xeve_tbl_itx_hip", "xeve_itrans_map_tbl_hip",
             "xevem_scaled_horizontal_sobel_filter_hip", "xevem_scaled_vertical_sobel_filter_hip", "xevem_equal_coeff_computer_hip", "xeve_tbl_intra_pred_ang_hip")
_refm = None


def ref_main_lib():
    global _refm
    if _refm is None and os.path.exists(REFM_SO):
        _refm = C.CDLL(REFM_SO)
    return _refm


def input_checksum():
    import zlib
    c = 0
    for case in mc_cases():
        c = zlib.crc32(case[6].tobytes(), zlib.crc32(np.array(case[:6] + case[7:], np.int64).tobytes(), c))
    for case in tx_cases():
        c = zlib.crc32(case[4].tobytes(), zlib.crc32(np.array(case[:4], np.int64).tobytes(), c))
    for case in ats_cases():
        c = zlib.crc32(case[6].tobytes(), zlib.crc32(np.array(case[:6], np.int64).tobytes(), c))
    for case in sobel_cases():
        c = zlib.crc32(case[3].tobytes(), zlib.crc32(np.array(case[:3] + case[4:], np.int64).tobytes(), c))
    for case in eq_cases():
        for a in case[3:]:
            c = zlib.crc32(a.tobytes(), c)
    for case in ang_cases():
        c = zlib.crc32(np.array(case[:6], np.int64).tobytes(), c)
        for a in case[6:]:
            c = zlib.crc32(a.tobytes(), c)
    return c


def check_golden(impl):
    z = np.load(GOLDEN)
    assert int(z["inputs_crc"]) == input_checksum(), "the seeded inputs differ from the ones the golden file was made on"
    g = z["out"]
    got = np.concatenate(run_all(impl))
    assert got.size == g.size
    assert np.array_equal(got, g)
    return got.size
