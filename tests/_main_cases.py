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


class TableMain:
    """any library exporting the five tables under `names` (the reference's C or SIMD variants, or libxeve_hip.so's *_hip tables)"""

    def __init__(self, L, names):
        self.L = L
        self.t = [(FN_MCM * 4).in_dll(L, names[k]) for k in range(3)]
        self.f = (FN_TX * 6).in_dll(L, names[3])
        self.i = (FN_TX * 6).in_dll(L, names[4])

    def mc(self, kind, fx, fy, plane, org, gx, gy, s_ref, sp, pred, w, h, bd):
        self.t[kind][int(fx) * 2 + int(fy)](ptr(plane, org), gx, gy, s_ref, sp, ptr(pred), w, h, bd)

    def tx(self, fwd, log2n, src, dst, shift, line):
        (self.f if fwd else self.i)[log2n - 1](ptr(src), ptr(dst), shift, line)


REF_NAMES = {
    "c": ("xevem_tbl_dmvr_mc_l", "xevem_tbl_dmvr_mc_c", "xevem_tbl_bl_mc_l", "xeve_tbl_tx", "xeve_tbl_itx"),
    "sse": ("xeve_tbl_dmvr_mc_l_sse", "xeve_tbl_dmvr_mc_c_sse", "xeve_tbl_bl_mc_l_sse", "xeve_tbl_tx", "xeve_tbl_itx"),
}
HIP_NAMES = ("xevem_tbl_dmvr_mc_l_hip", "xevem_tbl_dmvr_mc_c_hip", "xevem_tbl_bl_mc_l_hip", "xeve_tbl_tx_hip", "xeve_tbl_itx_hip")
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
    return c


def check_golden(impl):
    z = np.load(GOLDEN)
    assert int(z["inputs_crc"]) == input_checksum(), "the seeded inputs differ from the ones the golden file was made on"
    g = z["out"]
    got = np.concatenate(run_all(impl))
    assert got.size == g.size
    assert np.array_equal(got, g)
    return got.size
