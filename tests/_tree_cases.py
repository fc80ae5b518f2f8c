"""Seeded cases for the mode decision of I-picture CTUs (mode_analyze_lcu -> mode_coding_tree, src_base/xeve_mode.c:2007-2610): whole small pictures, coded CTU by
CTU in raster order from a clean state (maps empty, reconstruction grey), each CTU entering with the coder state the previous CTU's decision left.  Content is a
mix of smooth areas (where the early-termination rule of I pictures stops the split) and noise (where the tree goes down to 4x4); some pictures end in partial
CTUs.  The oracle (xo_mode_analyze_ctu_intra) is pinned beside the live encoder by tests/test_integration_ref.py::test_oracle_ctu_mode_decision_matches_the_live_encoder."""
import ctypes as C

import numpy as np

from _intra_cases import IntraParams
from _libs import SBAC_DTYPE, c_int, c_void_p, oracle, ptr
from _sbac_cases import make_states

CU_DEPTHS = 10


class TreeParams(C.Structure):  # xo_tree_params / xeve_hip_tree_params
    _fields_ = [("ip", IntraParams), ("pic_w", c_int), ("pic_h", c_int), ("log2_ctu", c_int), ("max_cu", c_int), ("min_cu", c_int), ("min_cuwh", c_int),
                ("slice_qp", c_int), ("slice_num", c_int), ("pad_", c_int)]


CTU_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("sbac", "<i4"), ("pic", "<i4")])
CTU_DATA_DTYPE = np.dtype([("split_mode", "i1", (CU_DEPTHS, 256)), ("pred_mode", "u1", (256,)), ("ipm", "i1", (2, 256)), ("depth", "i1", (256,)), ("nnz", "<i4", (3, 256)),
                           ("map_scu", "<u4", (256,)), ("map_cu_mode", "<u4", (256,)), ("coef", "<i2", (3, 4096)), ("reco", "<i2", (3, 4096)), ("mv", "<i2", (256, 2, 2)),
                           ("mvd", "<i2", (256, 2, 2)), ("refi", "i1", (256, 2)), ("mvp_idx", "u1", (256, 2))])
assert C.sizeof(TreeParams) == 136 and CTU_DATA_DTYPE.itemsize == 62976

# seed, pictures, w, h, bit depth, chroma_format_idc, log2 CTU, max_cu_intra, min_cu_intra, qp
CASES = [(3101, 3, 128, 64, 10, 1, 6, 32, 4, 37), (3102, 2, 72, 88, 10, 1, 6, 64, 4, 30), (3103, 2, 64, 64, 8, 0, 5, 32, 8, 42), (3104, 2, 96, 64, 10, 3, 6, 32, 4, 34),
         (3105, 2, 64, 32, 10, 1, 4, 16, 4, 27)]


def make_case(seed, npic, w, h, bd, idc, log2_ctu, max_cu, min_cu, qp8):
    r = np.random.default_rng(seed)
    maxv = (1 << bd) - 1
    ws, hs = (1 if idc in (1, 2) else 0), (1 if idc == 1 else 0)
    wc, hc = (w >> ws, h >> hs) if idc else (1, 1)
    org = [np.zeros((npic, h, w), np.int16), np.zeros((npic, hc, wc), np.int16), np.zeros((npic, hc, wc), np.int16)]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for p in range(npic):
        base = (maxv / 2) * (1 + 0.45 * np.sin(xx / (9.0 + 3 * p)) * np.cos(yy / 13.0) + 0.25 * np.sin((xx + 2 * yy) / 23.0))
        # noise in some 16x16 tiles, almost none in the others
        amp = np.kron(r.choice([0, 1, 12, 60], size=((h + 15) // 16, (w + 15) // 16)), np.ones((16, 16)))[:h, :w] * (1 << (bd - 8))
        luma = base + (r.random((h, w)) * 2 - 1) * amp
        org[0][p] = np.clip(luma, 0, maxv).astype(np.int16)
        if idc:
            for c in (1, 2):
                sub = luma[::(2 if hs else 1), ::(2 if ws else 1)] * (0.8 if c == 1 else 1.15)
                org[c][p] = np.clip(sub, 0, maxv).astype(np.int16)
    w_scu, h_scu = w // 4, h // 4
    P = TreeParams()
    qp = qp8 + 6 * (bd - 8)
    P.ip.w_scu, P.ip.h_scu, P.ip.slice_type, P.ip.chroma_format_idc, P.ip.bit_depth, P.ip.tool_iqt, P.ip.constrained_intra_pred = w_scu, h_scu, 2, idc, bd, 0, 0
    P.ip.qp[0], P.ip.qp[1], P.ip.qp[2] = qp, qp - 1, qp - 2
    lam = 0.57 * 2.0 ** ((qp8 - 12) / 3.0)
    P.ip.lambda_[0] = lam
    P.ip.dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = 2.0 ** (1 / 3.0), 2.0 ** (2 / 3.0)
    P.ip.lambda_[1], P.ip.lambda_[2] = lam / P.ip.dist_chroma_weight[0], lam / P.ip.dist_chroma_weight[1]
    P.ip.sqrt_lambda0 = float(np.sqrt(lam))
    P.pic_w, P.pic_h, P.log2_ctu, P.max_cu, P.min_cu, P.min_cuwh, P.slice_qp, P.slice_num = w, h, log2_ctu, max_cu, min_cu, 4, qp, 0
    entry = make_states(r, npic)
    grey = 1 << (bd - 1)
    mod = [np.full_like(a, grey) for a in org]
    maps = dict(scu=np.zeros((npic, h_scu * w_scu), np.uint32), ipm=np.zeros((npic, h_scu * w_scu), np.int8), tidx=np.zeros((npic, h_scu * w_scu), np.uint8),
                cu_mode=np.zeros((npic, h_scu * w_scu), np.uint32))
    ctu = 1 << log2_ctu
    order = [(x, y) for y in range(0, h, ctu) for x in range(0, w, ctu)]
    return dict(org=org, mod=mod, maps=maps, P=P, entry=entry, npic=npic, w=w, h=h, idc=idc, order=order)


def oracle_tree():
    L = oracle()
    L.xo_mode_analyze_ctu_intra.restype = C.c_double
    L.xo_mode_analyze_ctu_intra.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 5 + [C.POINTER(TreeParams), c_int, c_int, c_void_p, c_void_p]
    return L


def run_oracle_picture(c, p):
    """codes picture p of the case CTU by CTU; mod / maps of the case are updated in place.  Returns per CTU (ctu data record, next_best record, cost)"""
    O = oracle_tree()
    org = (c_void_p * 3)(*[a[p].ctypes.data for a in c["org"]])
    mod = (c_void_p * 3)(*[a[p].ctypes.data for a in c["mod"]])
    m = c["maps"]
    state = c["entry"][p:p + 1].copy()
    out = []
    for (x, y) in c["order"]:
        d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
        cost = O.xo_mode_analyze_ctu_intra(org, c["org"][0].shape[2], c["org"][1].shape[2], mod, c["mod"][0].shape[2], c["mod"][1].shape[2], ptr(m["scu"][p]),
                                           ptr(m["ipm"][p]), ptr(m["tidx"][p]), ptr(m["cu_mode"][p]), ptr(state), C.byref(c["P"]), x, y, ptr(d), ptr(nb))
        out.append((d, nb, cost))
        state = nb.copy()
    return out


# ---- P / B slices: one picture with its reference pictures; the CTUs coded in raster order from clean maps -------------------------------------------------------
from _inter_cases import make_inter_params, make_inter_picture  # noqa: E402
from _libs import InterParams  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402


class TreeInter(C.Structure):  # xo_tree_inter
    _fields_ = [("refp", c_void_p), ("s_ref_l", c_int), ("s_ref_c", c_int), ("ipar", InterParams), ("map_mv", c_void_p), ("map_refi", c_void_p), ("col0", c_void_p),
                ("col1", c_void_p), ("ecu_depth", c_int), ("pad_", c_int)]


# seed, w, h, bit depth, chroma_format_idc, slice type (0 B, 1 P), reference pictures per list, odd POC (the early CU termination starts two depths higher), skip_th
INTER_CASES = [(4101, 128, 64, 10, 1, 1, 1, 0, 0.0), (4102, 128, 64, 10, 1, 0, 2, 1, 0.0), (4103, 72, 88, 10, 1, 0, 1, 0, 0.0), (4104, 64, 64, 8, 0, 1, 2, 1, 0.0),
               (4105, 128, 128, 10, 1, 0, 2, 0, 6.0)]


def make_inter_case(seed, w, h, bd, idc, slice_type, nref, odd_poc, skip_th):
    r = np.random.default_rng(seed)
    refs, org = make_inter_picture(r, w, h, bd, nref, idc, slice_type)
    # local motion and new content, so that the tree splits and intra CUs win somewhere: rectangles of the original replaced by a differently shifted copy of
    # itself, by noise, or by a flat patch (chroma follows)
    ws_, hs_ = refs["ws"], refs["hs"]
    views = [org[0][refs["org_l"] // refs["s_l"]:, :][:h, refs["org_l"] % refs["s_l"]:][:, :w]]
    if idc:
        views += [org[k][refs["org_c"] // refs["s_c"]:, :][:h >> hs_, refs["org_c"] % refs["s_c"]:][:, :w >> ws_] for k in (1, 2)]
    for _ in range(int(r.integers(4, 9))):
        bw, bh = int(r.choice([8, 16, 32])), int(r.choice([8, 16, 32]))
        bx, by = int(r.integers(0, (w - bw) // 8 + 1)) * 8, int(r.integers(0, (h - bh) // 8 + 1)) * 8
        kind = int(r.integers(0, 3))
        dx, dy = int(r.integers(-3, 4)) * 2, int(r.integers(-3, 4)) * 2
        for k, v in enumerate(views):
            sx, sy = (ws_, hs_) if k else (0, 0)
            x0, y0, x1, y1 = bx >> sx, by >> sy, (bx + bw) >> sx, (by + bh) >> sy
            if kind == 0:
                v[y0:y1, x0:x1] = np.roll(v, (dy >> sy, dx >> sx), axis=(0, 1))[y0:y1, x0:x1]
            elif kind == 1:
                v[y0:y1, x0:x1] = r.integers(0, 1 << bd, size=(y1 - y0, x1 - x0))
            else:
                v[y0:y1, x0:x1] = int(r.integers(0, 1 << bd))
    ipar = make_inter_params(r, 6, w, h, bd, nref, idc, slice_type, refs, skip_th)
    rp = ipar.rdo
    P = TreeParams()
    P.ip.w_scu, P.ip.h_scu, P.ip.slice_type, P.ip.chroma_format_idc, P.ip.bit_depth, P.ip.tool_iqt, P.ip.constrained_intra_pred = w // 4, h // 4, slice_type, idc, bd, 0, 0
    for c in range(3):
        P.ip.qp[c], P.ip.lambda_[c] = rp.qp[c], rp.lambda_[c]
    P.ip.dist_chroma_weight[0], P.ip.dist_chroma_weight[1], P.ip.sqrt_lambda0 = rp.dist_chroma_weight[0], rp.dist_chroma_weight[1], float(np.sqrt(rp.lambda_[0]))
    P.pic_w, P.pic_h, P.log2_ctu, P.max_cu, P.min_cu, P.min_cuwh, P.slice_qp, P.slice_num = w, h, 6, 64, 8, 4, rp.qp[0] - 6 * (bd - 8), 0
    nscu = (w // 4) * (h // 4)
    grey = 1 << (bd - 1)
    ws, hs = refs["ws"], refs["hs"]
    mod = [np.full((h, w), grey, np.int16), np.full((max(1, h >> hs), max(1, w >> ws)), grey, np.int16), np.full((max(1, h >> hs), max(1, w >> ws)), grey, np.int16)]
    maps = dict(scu=np.zeros(nscu, np.uint32), ipm=np.zeros(nscu, np.int8), tidx=np.zeros(nscu, np.uint8), cu_mode=np.zeros(nscu, np.uint32),
                mv=np.zeros((nscu, 2, 2), np.int16), refi=np.full((nscu, 2), -1, np.int8))
    col = [r.integers(-24, 25, size=(nscu, 2, 2)).astype(np.int16) for _ in range(2)]
    entry = make_states(r, 1)
    order = [(x, y) for y in range(0, h, 64) for x in range(0, w, 64)]
    return dict(refs=refs, org=org, mod=mod, maps=maps, col=col, P=P, ipar=ipar, entry=entry, idc=idc, w=w, h=h, order=order, ecu_depth=2 if odd_poc else 4, slice_type=slice_type)


def oracle_tree_any():
    L = oracle()
    L.xo_mode_analyze_ctu.restype = C.c_double
    L.xo_mode_analyze_ctu.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 5 + [C.POINTER(TreeParams), C.POINTER(TreeInter), c_int, c_int, c_void_p, c_void_p]
    return L


def run_oracle_inter_picture(c):
    """codes the picture of the case CTU by CTU; mod / maps of the case are updated in place.  Returns per CTU (ctu data record, next_best record, cost)"""
    O = oracle_tree_any()
    refs, org = c["refs"], c["org"]
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    I = TreeInter()
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, refs["s_l"], refs["s_c"], c["ipar"]
    m = c["maps"]
    I.map_mv, I.map_refi, I.col0, I.col1, I.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, c["col"][0].ctypes.data, c["col"][1].ctypes.data, c["ecu_depth"]
    orgp = (c_void_p * 3)(int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"])
    modp = (c_void_p * 3)(*[a.ctypes.data for a in c["mod"]])
    state = c["entry"][0:1].copy()
    out = []
    for (x, y) in c["order"]:
        d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
        cost = O.xo_mode_analyze_ctu(orgp, refs["s_l"], refs["s_c"], modp, c["mod"][0].shape[1], c["mod"][1].shape[1], ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]),
                                     ptr(m["cu_mode"]), ptr(state), C.byref(c["P"]), C.byref(I), x, y, ptr(d), ptr(nb))
        out.append((d, nb, cost))
        state = nb.copy()
    return out
