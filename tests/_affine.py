"""Helpers of the affine motion-compensation tests (Main profile): seeded cases, the oracle's restatement (oracle/xeve_oracle.c xo_affine_mc) and the
reference's own xeve_affine_mc called in place through oracle/ref_affine_driver.c (build container only); goldens: tests/golden/make_affine_golden.py."""
import ctypes as C
import os

import numpy as np

from _libs import ORACLE_DIR, ROOT, oracle

GOLDEN = os.path.join(ROOT, "tests", "golden", "affine_v1.npz")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_affine.so")
PAD, BD, PIC_W, PIC_H = 160, 10, 256, 192
REFPIC = np.dtype([("y", "<u8"), ("u", "<u8"), ("v", "<u8"), ("poc", "<i4"), ("pad_", "<i4")])  # xo_refpic / xeve_hip_refpic
JOB = np.dtype([("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 3, 2)), ("refi", "i1", (2,)), ("vertex_num", "i1"), ("pad_", "i1")])  # xo_affine_job / xeve_hip_affine_job
assert REFPIC.itemsize == 32 and JOB.itemsize == 36


def ref_pictures(seed, n=2):
    """n reference pictures per list: [refi][list] -> (Y, U, V) planes with PAD / PAD/2 samples of margin; smooth texture + noise, different per picture"""
    pics = []
    for r in range(n):
        row = []
        for l in range(2):
            g = np.random.default_rng(seed * 100 + r * 2 + l)
            comps = []
            for c in range(3):
                w, h, p = (PIC_W, PIC_H, PAD) if c == 0 else (PIC_W // 2, PIC_H // 2, PAD // 2)
                yy, xx = np.mgrid[0:h + 2 * p, 0:w + 2 * p]
                a = 512 + 280 * np.sin(xx / (6.0 + c + r) + yy / (9.0 + l)) + 150 * np.cos(yy / 4.0 - xx / 13.0) + g.integers(-90, 91, size=xx.shape)
                comps.append(np.ascontiguousarray(np.clip(a, 0, 1023).astype(np.int16)))
            row.append(comps)
        pics.append(row)
    return pics


def plane_ptr(a, c):
    p = PAD if c == 0 else PAD // 2
    return a.ctypes.data + 2 * (p * a.shape[1] + p)


def make_jobs(w, h, seed, n=24):
    """n CUs of w x h: positions inside the picture and at / across its borders, uni- and bi-prediction, both models, control-point vectors whose differences span
    'no change' .. 'beyond the 4x4 limit' (sub-blocks of the CU's size, 32, 16, 8, and the enhanced interpolation filter), large base vectors that hit the clip ranges"""
    g = np.random.default_rng(seed)
    jobs = np.zeros(n, JOB)
    for i in range(n):
        edge = i % 6
        x = [int(g.integers(0, (PIC_W - w) // 4 + 1)) * 4, 0, PIC_W - w, int(g.integers(0, (PIC_W - w) // 4 + 1)) * 4, PIC_W - w, 0][edge]
        y = [int(g.integers(0, (PIC_H - h) // 4 + 1)) * 4, int(g.integers(0, (PIC_H - h) // 4 + 1)) * 4, 0, PIC_H - h, PIC_H - h, 0][edge]
        jobs[i]["x"], jobs[i]["y"] = x, y
        kind = i % 3
        jobs[i]["refi"] = [(0, -1), (-1, int(g.integers(0, 2))), (int(g.integers(0, 2)), 0)][kind]
        jobs[i]["vertex_num"] = 2 + (i // 3) % 2
        spread = [0, 1, 2, 3, 4, 6, 12, 40][(i // 2) % 8] * max(w, h) // 32 + (1 if i % 5 == 0 else 0)  # quarter-pel difference between the control points
        big = 600 if i % 7 == 3 else 40
        for l in range(2):
            base = g.integers(-big, big + 1, size=2)
            for v in range(3):
                jobs[i]["mv"][l][v] = base + (g.integers(-spread, spread + 1, size=2) if v else 0)
            if i % 8 == 7:  # a horizontal zoom of 0.8 .. 1.1 with nothing else: the 4x4 block's bounding box exceeds the bandwidth limit while the fetched-lines test
                jobs[i]["mv"][l][1] = base + (int((3.3 + 0.3 * l + 0.1 * (i // 8)) * w), 0)  # passes -- the enhanced filter WITH the range around the centre vector
                jobs[i]["mv"][l][2] = base
                jobs[i]["vertex_num"] = 3
    return jobs


SIZES = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 32), (64, 16), (32, 128)]


def refp_table(pics):
    t = np.zeros(len(pics) * 2, REFPIC)
    for r, row in enumerate(pics):
        for l, comps in enumerate(row):
            t[r * 2 + l] = (plane_ptr(comps[0], 0), plane_ptr(comps[1], 1), plane_ptr(comps[2], 2), 8 * r + l, 0)
    return t


def strides(pics):
    return pics[0][0][0].shape[1], pics[0][0][1].shape[1]


class OracleAffine:
    name = "oracle"

    def __init__(self):
        self.L = oracle()
        self.L.xo_affine_mc.restype = None
        self.L.xo_affine_mc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def run(self, pics, jobs, w, h):
        """-> (Y [n][h][w], U, V, path [n][3])"""
        n, t = len(jobs), refp_table(pics)
        s_l, s_c = strides(pics)
        Y, U, V, path = np.zeros((n, h, w), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, 3), np.int32)
        for i in range(n):
            self.L.xo_affine_mc(t.ctypes.data, s_l, s_c, PIC_W, PIC_H, jobs[i:i + 1].ctypes.data, w, h, BD, Y[i].ctypes.data, U[i].ctypes.data, V[i].ctypes.data, path[i].ctypes.data)
        return Y, U, V, path


class RefAffine:
    name = "reference"

    def __init__(self):
        self.L = C.CDLL(REF_SO)
        self.L.refdrv_affine_mc.restype = C.c_int

    def run(self, pics, jobs, w, h):
        n = len(jobs)
        s_l, s_c = strides(pics)
        planes = (C.c_void_p * (len(pics) * 2 * 3))()
        for r, row in enumerate(pics):
            for l, comps in enumerate(row):
                for c in range(3):
                    planes[(r * 2 + l) * 3 + c] = plane_ptr(comps[c], c)
        Y, U, V, path = np.zeros((n, h, w), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, 3), np.int32)
        for i in range(n):
            j = jobs[i]
            refi = (C.c_int8 * 2)(*[int(v) for v in j["refi"]])
            mv = np.ascontiguousarray(j["mv"])
            self.L.refdrv_affine_mc(C.c_int(int(j["x"])), C.c_int(int(j["y"])), C.c_int(PIC_W), C.c_int(PIC_H), C.c_int(w), C.c_int(h), refi, C.c_void_p(mv.ctypes.data), planes,
                                    C.c_int(s_l), C.c_int(s_c), C.c_int(int(j["vertex_num"])), C.c_int(BD), C.c_void_p(Y[i].ctypes.data), C.c_void_p(U[i].ctypes.data),
                                    C.c_void_p(V[i].ctypes.data), C.c_void_p(path[i].ctypes.data))
        return Y, U, V, path


def digests(Y, U, V):
    """[n][16] uint8: md5 of every job's three prediction planes (what the golden file holds: a 128 x 128 CU's planes are 48 KB)"""
    import hashlib

    return np.stack([np.frombuffer(hashlib.md5(Y[i].tobytes() + U[i].tobytes() + V[i].tobytes()).digest(), np.uint8) for i in range(len(Y))])


HOST_SRC = os.path.join(ROOT, "tests", "native", "affine_host.cpp")
HOST_HDR = os.path.join(ROOT, "xeve_amd", "csrc", "affine_core.h")
HOST_OUT = os.path.join(ROOT, "tests", "native", "build", "libaffine_host.so")


class HostAffine(OracleAffine):
    """affine_core.h compiled by g++ (tests/native/affine_host.cpp): the kernel's per-CU set-up and per-sample functions"""
    name = "affine_core.h on the host"

    def __init__(self):
        import subprocess

        OracleAffine.__init__(self)
        if not os.path.exists(HOST_OUT) or os.path.getmtime(HOST_OUT) < max(os.path.getmtime(HOST_SRC), os.path.getmtime(HOST_HDR)):
            os.makedirs(os.path.dirname(HOST_OUT), exist_ok=True)
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-o", HOST_OUT, HOST_SRC], check=True)
        self.H = C.CDLL(HOST_OUT)
        self.H.xa_host_affine_mc.restype = None
        self.cl = (C.c_int16 * 128).in_dll(self.L, "xom_mc_l_coeff")  # the Main filters (pinned against the reference's tables by tests/test_main_oracle_vs_ref.py)
        self.cc = (C.c_int16 * 128).in_dll(self.L, "xom_mc_c_coeff")

    def run(self, pics, jobs, w, h):
        n, t = len(jobs), refp_table(pics)
        s_l, s_c = strides(pics)
        Y, U, V, path = np.zeros((n, h, w), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, h // 2, w // 2), np.int16), np.zeros((n, 3), np.int32)
        for i in range(n):
            self.H.xa_host_affine_mc(C.c_void_p(t.ctypes.data), C.c_int(s_l), C.c_int(s_c), C.c_int(PIC_W), C.c_int(PIC_H), C.c_void_p(jobs[i:i + 1].ctypes.data), C.c_int(w), C.c_int(h),
                                     C.c_int(BD), self.cl, self.cc, C.c_void_p(Y[i].ctypes.data), C.c_void_p(U[i].ctypes.data), C.c_void_p(V[i].ctypes.data),
                                     C.c_void_p(path[i].ctypes.data))
        return Y, U, V, path
