"""Test infrastructure: builds and loads the HOST side of xeve_amd/csrc/cu_lane.h (the lane-serial intra analysis of a small CU that libxeve_hip.so runs on the
device, every function __host__ __device__) so that the CPU suite can compare it bit for bit with the oracle.  hipcc --cuda-host-only; nothing of this is linked
into the product library."""
import ctypes as C
import os
import subprocess

import numpy as np

from _libs import ROOT, c_int, c_void_p, oracle

SRC = os.path.join(ROOT, "tests", "native", "cu_lane_host.cpp")
HDR = os.path.join(ROOT, "xeve_amd", "csrc", "cu_lane.h")
HDR2 = os.path.join(ROOT, "xeve_amd", "csrc", "eco_lane.h")
OUT = os.path.join(ROOT, "tests", "native", "build", "libcu_lane_host.so")
HIPCC = "/opt/rocm/bin/hipcc"


class LaneParams(C.Structure):  # xl::Params
    _fields_ = [("idc", c_int), ("bd", c_int), ("slice_type", c_int), ("cip", c_int), ("w_scu", c_int), ("h_scu", c_int), ("s_org_l", c_int), ("s_org_c", c_int),
                ("s_mod_l", c_int), ("s_mod_c", c_int), ("qp", c_int * 3), ("q_scale", c_int * 3), ("dq_scale", c_int * 3), ("err_scale", C.c_int64 * 3),
                ("lambda_", C.c_double * 3), ("sqrt_lambda0", C.c_double), ("wgt", C.c_double * 2), ("entropy", c_void_p)]


_lib = None
_entropy = None
QUANT_SCALE = [26214, 23302, 20560, 18396, 16384, 14764]
DQ_SCALE = [40, 45, 51, 57, 64, 71]


def available():
    return os.path.exists(HIPCC)


def lane():
    global _lib
    if _lib is None:
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.run([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-o", OUT, SRC], check=True)
        _lib = C.CDLL(OUT)
        _lib.xl_host_intra_cu.restype = None
        _lib.xl_host_intra_cu.argtypes = [c_int, C.POINTER(LaneParams)] + [c_void_p] * 15
        assert _lib.xl_host_sizeof_params() == C.sizeof(LaneParams)
        _lib.xl_host_eco_ctu.restype = c_int
        _lib.xl_host_eco_ctu.argtypes = [c_int] * 8 + [c_void_p] * 9 + [c_int, c_int, c_void_p, c_int]
        _lib.xl_host_eco_tile_end.restype = c_int
        _lib.xl_host_eco_tile_end.argtypes = [c_void_p, c_void_p, c_int]
    return _lib


def entropy_table():
    global _entropy
    if _entropy is None:
        O = oracle()
        O.xo_entropy_bits.restype = C.c_int32
        O.xo_entropy_bits.argtypes = [c_int]
        _entropy = np.array([O.xo_entropy_bits(i) for i in range(1026)], np.int32)
    return _entropy


def lane_params(ip, log2, s_org_l, s_org_c, s_mod_l, s_mod_c):
    """xl::Params of one CU size from the intra parameters (IntraParams of _intra_cases / lib.IntraParams)"""
    O = oracle()
    O.xo_err_scale.restype = C.c_int64
    O.xo_err_scale.argtypes = [c_int] * 4
    P = LaneParams()
    idc, bd = ip.chroma_format_idc, ip.bit_depth
    P.idc, P.bd, P.slice_type, P.cip, P.w_scu, P.h_scu = idc, bd, ip.slice_type, ip.constrained_intra_pred, ip.w_scu, ip.h_scu
    P.s_org_l, P.s_org_c, P.s_mod_l, P.s_mod_c = s_org_l, s_org_c, s_mod_l, s_mod_c
    lc = log2 - (1 if idc in (1, 2) else 0)
    for c in range(3):
        q = ip.qp[c]
        P.qp[c], P.q_scale[c], P.dq_scale[c] = q, QUANT_SCALE[q % 6], DQ_SCALE[q % 6] << (q // 6)
        P.err_scale[c] = O.xo_err_scale(q % 6, log2 if c == 0 else lc, bd, 0)
        P.lambda_[c] = ip.lambda_[c]
    P.sqrt_lambda0, P.wgt[0], P.wgt[1] = ip.sqrt_lambda0, ip.dist_chroma_weight[0], ip.dist_chroma_weight[1]
    P.entropy = entropy_table().ctypes.data
    return P


_scans = None


def scans():
    """zig-zag scans of the 16x16, 32x32, 64x64 blocks (xeve_tbl_scan) from the oracle's xo_zigzag"""
    global _scans
    if _scans is None:
        O = oracle()
        O.xo_zigzag.restype = None
        O.xo_zigzag.argtypes = [c_int, c_int, c_void_p]
        _scans = []
        for l in (4, 5, 6):
            a = np.zeros(1 << (2 * l), np.uint16)
            O.xo_zigzag(l, l, a.ctypes.data)
            _scans.append(a)
    return _scans
