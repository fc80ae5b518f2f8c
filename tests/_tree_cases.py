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
                           ("map_scu", "<u4", (256,)), ("map_cu_mode", "<u4", (256,)), ("coef", "<i2", (3, 4096)), ("reco", "<i2", (3, 4096))])
assert C.sizeof(TreeParams) == 136 and CTU_DATA_DTYPE.itemsize == 57856

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
