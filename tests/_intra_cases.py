"""Seeded cases for the intra analysis of a CU (pintra_analyze_cu, src_base/xeve_pintra.c:544-698): an original picture with smooth content plus noise, a
"mode picture" (the reconstruction so far) that resembles it, and 4x4-unit maps in which the units before the CU in coding order are coded -- some of them
intra, some in another tile -- so that every availability rule of xeve_get_nbr / xeve_get_mpm is exercised."""
import ctypes as C
import os

import numpy as np

from _libs import ORACLE_DIR, SBAC_DTYPE, c_int, c_void_p, oracle, ptr
from _sbac_cases import make_states

REF_INTRA_SO = os.path.join(ORACLE_DIR, "_ref", "libref_intra.so")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "intra_v1.npz")


class IntraParams(C.Structure):  # xo_intra_params / xeve_hip_intra_params
    _fields_ = [("log2_cuw", c_int), ("log2_cuh", c_int), ("w_scu", c_int), ("h_scu", c_int), ("slice_type", c_int), ("chroma_format_idc", c_int),
                ("bit_depth", c_int), ("tool_iqt", c_int), ("constrained_intra_pred", c_int), ("qp", c_int * 3), ("lambda_", C.c_double * 3),
                ("sqrt_lambda0", C.c_double), ("dist_chroma_weight", C.c_double * 2)]


INTRA_JOB_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("inter_satd", "<u4"), ("sbac", "<i4"), ("pic", "<i4"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1"), ("pad_", "u1", (2,))])
INTRA_RESULT_DTYPE = np.dtype([("cost", "<f8"), ("dist_cu", "<i4"), ("nnz", "<i4", (3,)), ("pred_cnt", "<i4"), ("ipm", "i1", (2,)), ("pad_", "i1", (2,))])
assert C.sizeof(IntraParams) == 96 and INTRA_JOB_DTYPE.itemsize == 24 and INTRA_RESULT_DTYPE.itemsize == 32

# seed, w, h, bit depth, chroma_format_idc, slice type (0 B, 1 P, 2 I), log2 CU size, constrained intra prediction, tiles
CASES = [(1101, 128, 96, 10, 1, 2, 3, 0, 0), (1102, 128, 96, 10, 1, 2, 4, 0, 0), (1103, 128, 128, 10, 1, 2, 5, 0, 0), (1104, 128, 128, 10, 1, 2, 6, 0, 0),
         (1105, 128, 96, 10, 1, 0, 3, 0, 0), (1106, 128, 96, 10, 1, 0, 4, 1, 0), (1107, 96, 64, 8, 1, 1, 4, 0, 1), (1108, 64, 64, 10, 0, 2, 3, 0, 0),
         (1109, 128, 96, 10, 1, 2, 2, 0, 0), (1110, 128, 128, 10, 1, 0, 5, 1, 1), (1111, 96, 96, 12, 3, 2, 4, 0, 0)]
N_JOBS = 16


def make_case(seed, w, h, bd, idc, slice_type, lw, cip, tiles):
    r = np.random.default_rng(seed)
    maxv = (1 << bd) - 1
    ws, hs = (1 if idc in (1, 2) else 0), (1 if idc == 1 else 0)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = (maxv / 2) * (1 + 0.5 * np.sin(xx / 11.0) * np.cos(yy / 7.0) + 0.3 * np.sin((xx + 2 * yy) / 17.0))
    amp = int(r.choice([2, 8, 40])) << (bd - 8)
    org, mod = [], []
    for c in range(3):
        b = base if c == 0 else base[::(2 if hs else 1), ::(2 if ws else 1)] * (0.9 if c == 1 else 1.1)
        org.append(np.clip(b + r.integers(-amp, amp + 1, size=b.shape), 0, maxv).astype(np.int16))
        mod.append(np.clip(b + r.integers(-amp, amp + 1, size=b.shape), 0, maxv).astype(np.int16))
    w_scu, h_scu = w // 4, h // 4
    st = make_states(r, 6)
    P = IntraParams()
    P.log2_cuw = P.log2_cuh = lw
    P.w_scu, P.h_scu, P.slice_type, P.chroma_format_idc, P.bit_depth, P.tool_iqt, P.constrained_intra_pred = w_scu, h_scu, slice_type, idc, bd, 0, cip
    qp = int(r.integers(22, 46)) + 6 * (bd - 8)
    dq = int(r.integers(-3, 4))
    P.qp[0], P.qp[1], P.qp[2] = qp, max(0, qp + dq), max(0, qp + dq - 1)
    lam = 0.57 * 2.0 ** ((qp - 6 * (bd - 8) - 12) / 3.0) * (0.8 + 0.4 * float(r.random()))
    P.lambda_[0] = lam
    P.dist_chroma_weight[0], P.dist_chroma_weight[1] = 2.0 ** (-dq / 3.0), 2.0 ** ((1 - dq) / 3.0)
    P.lambda_[1], P.lambda_[2] = lam / P.dist_chroma_weight[0], lam / P.dist_chroma_weight[1]
    P.sqrt_lambda0 = float(np.sqrt(lam))
    cu = 1 << lw
    jobs = np.zeros(N_JOBS, INTRA_JOB_DTYPE)
    jobs["x"] = r.integers(0, (w - cu) // cu + 1, size=N_JOBS) * cu
    jobs["y"] = r.integers(0, (h - cu) // cu + 1, size=N_JOBS) * cu
    jobs["x"][0], jobs["y"][0] = 0, 0  # the picture corner: nothing available
    jobs["x"][1], jobs["y"][1] = (w - cu) // cu * cu, 0
    jobs["x"][2], jobs["y"][2] = 0, (h - cu) // cu * cu
    jobs["x"][3], jobs["y"][3] = (w - cu) // cu * cu, (h - cu) // cu * cu
    u = r.random(N_JOBS)
    jobs["inter_satd"] = np.where((slice_type == 2) | (u < 0.3), 0xFFFFFFFF, (r.integers(1, 40, size=N_JOBS) * cu * cu * (1 << (bd - 8)) // 4)).astype(np.uint32)
    jobs["sbac"] = r.integers(0, len(st), size=N_JOBS)
    jobs["ctx_skip"], jobs["ctx_pred_mode"] = r.integers(0, 2, size=N_JOBS), r.integers(0, 3, size=N_JOBS)
    # per job its own maps: the units before the CU in raster order of CU rows are coded (with holes), the rest not
    maps = []
    for i in range(N_JOBS):
        m = np.zeros(h_scu * w_scu, np.uint32)
        x_scu, y_scu, n = int(jobs["x"][i]) // 4, int(jobs["y"][i]) // 4, cu // 4
        yy_, xx_ = np.divmod(np.arange(h_scu * w_scu), w_scu)
        before = (yy_ < y_scu) | ((yy_ < y_scu + n + int(r.integers(0, 2 * n + 1))) & (xx_ < x_scu))  # rows above; a varying depth of the column to the left
        coded = before & (r.random(m.size) < 0.93)
        intra = coded & (r.random(m.size) < (0.5 if slice_type != 2 else 1.0))
        m |= (coded.astype(np.uint32) << 31) | (intra.astype(np.uint32) << 15)
        ipm = r.integers(0, 5, size=m.size).astype(np.int8)
        tidx = np.zeros(m.size, np.uint8)
        if tiles:
            tidx[(xx_ >= w_scu // 2)] = 1
        maps.append((m, ipm, tidx))
    return dict(org=org, mod=mod, states=st, P=P, jobs=jobs, maps=maps, idc=idc, lw=lw, w=w, h=h, n0=cu * cu, n1=(cu * cu) >> (ws + hs) if idc else 0)


_ARGT = [c_void_p] * 3 + [c_int, c_int] + [c_void_p] * 3 + [c_int, c_int] + [c_void_p] * 4 + [C.POINTER(IntraParams)] + [c_void_p] * 9


def oracle_intra():
    L = oracle()
    L.xo_pintra_analyze_cu.restype = None
    L.xo_pintra_analyze_cu.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 4 + [C.POINTER(IntraParams)] + [c_void_p] * 9
    L.xo_get_nbr.restype = None
    L.xo_get_nbr.argtypes = [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]
    L.xo_ipred.restype = None
    L.xo_ipred.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]
    return L


_refi = None


def ref_intra():
    global _refi
    if _refi is None and os.path.exists(REF_INTRA_SO):
        L = C.CDLL(REF_INTRA_SO)
        L.refdrv_pintra_analyze_cu.restype = None
        L.refdrv_pintra_analyze_cu.argtypes = _ARGT
        L.refdrv_get_nbr.restype = None
        L.refdrv_get_nbr.argtypes = [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p, c_int]
        L.refdrv_ipred.restype = None
        L.refdrv_ipred.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]
        _refi = L
    return _refi


def outputs(c):
    return (np.zeros(1, INTRA_RESULT_DTYPE), [np.zeros(c["n0"], np.int16), np.zeros(max(c["n1"], 1), np.int16), np.zeros(max(c["n1"], 1), np.int16)],
            [np.zeros(c["n0"], np.int16), np.zeros(max(c["n1"], 1), np.int16), np.zeros(max(c["n1"], 1), np.int16)], np.zeros(1, SBAC_DTYPE))


def run_oracle(c, i):
    O = oracle_intra()
    res, co, rc, best = outputs(c)
    org = (c_void_p * 3)(*[p.ctypes.data for p in c["org"]])
    mod = (c_void_p * 3)(*[p.ctypes.data for p in c["mod"]])
    m, ipm, tidx = c["maps"][i]
    O.xo_pintra_analyze_cu(org, c["org"][0].shape[1], c["org"][1].shape[1], mod, c["mod"][0].shape[1], c["mod"][1].shape[1], ptr(m), ptr(ipm), ptr(tidx),
                           ptr(c["states"]), C.byref(c["P"]), ptr(c["jobs"][i:i + 1]), ptr(res), ptr(co[0]), ptr(co[1]), ptr(co[2]), ptr(rc[0]), ptr(rc[1]), ptr(rc[2]),
                           ptr(best))
    return res, co, rc, best


def run_ref(c, i):
    R = ref_intra()
    res, co, rc, best = outputs(c)
    m, ipm, tidx = (a.copy() for a in c["maps"][i])
    o, md = [p.copy() for p in c["org"]], [p.copy() for p in c["mod"]]
    R.refdrv_pintra_analyze_cu(ptr(o[0]), ptr(o[1]), ptr(o[2]), o[0].shape[1], o[1].shape[1], ptr(md[0]), ptr(md[1]), ptr(md[2]), md[0].shape[1], md[1].shape[1],
                               ptr(m), ptr(ipm), ptr(tidx), ptr(c["states"]), C.byref(c["P"]), ptr(c["jobs"][i:i + 1]), ptr(res), ptr(co[0]), ptr(co[1]), ptr(co[2]),
                               ptr(rc[0]), ptr(rc[1]), ptr(rc[2]), ptr(best))
    return res, co, rc, best


def same(a, b, idc, what):
    ra, ca, ka, ba = a
    rb, cb, kb, bb = b
    for f in ("cost", "dist_cu", "nnz", "ipm"):
        assert ra[f].tobytes() == rb[f].tobytes(), (what, f, ra, rb)
    for k in range(3 if idc else 1):
        assert np.array_equal(ca[k], cb[k]), (what, "coef", k)
        assert np.array_equal(ka[k], kb[k]), (what, "rec", k)
    assert ba.tobytes() == bb.tobytes(), (what, "state")


def input_checksum(c):
    import zlib
    v = 0
    for a in c["org"] + c["mod"] + [c["states"].view(np.uint8), c["jobs"].view(np.uint8), np.frombuffer(bytes(c["P"]), np.uint8)] + [x for m in c["maps"] for x in m]:
        v = zlib.crc32(np.ascontiguousarray(a).tobytes(), v)
    return v


def golden():
    """(case dict, per-job expected outputs) of tests/golden/intra_v1.npz"""
    g = np.load(GOLDEN)
    for k, case in enumerate(CASES):
        c = make_case(*case)
        assert int(g["in_crc%d" % k]) == input_checksum(c), "the seeded inputs differ from the ones the golden file was made on"
        res = np.ascontiguousarray(g["res%d" % k]).view(INTRA_RESULT_DTYPE)
        best = np.ascontiguousarray(g["best%d" % k]).view(SBAC_DTYPE)
        exp = [(res[i:i + 1], [g["coef%d_%d" % (k, j)][i] for j in range(3)], [g["rec%d_%d" % (k, j)][i] for j in range(3)], best[i:i + 1]) for i in range(N_JOBS)]
        yield case, c, exp
