"""Helpers of the affine gradient-search tests (Main profile, SURVEY.md 8(f)4: "affine MC + gradient ME"): seeded cases, the oracle's restatement
(oracle/xeve_oracle.c xo_affine_me_gradient) and the reference's own static pinter_affine_me_gradient called in place through oracle/ref_affine_me_driver.c (build container
only); goldens: tests/golden/make_affine_me_golden.py."""
import ctypes as C
import os

import numpy as np

import _affine as A
from _libs import ORACLE_DIR, ROOT, oracle

GOLDEN = os.path.join(ROOT, "tests", "golden", "affine_me_v1.npz")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_affine_me.so")
PAD, BD, PIC_W, PIC_H = A.PAD, A.BD, A.PIC_W, A.PIC_H
# xo_affine_me_job / xeve_hip_affine_me_job
JOB = np.dtype([("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (3, 2)), ("mv", "<i2", (3, 2)), ("refi", "i1"), ("list", "i1"), ("bi", "i1"), ("vertex_num", "i1"), ("mot_bits_other", "<i4"),
                ("cost", "<u4")])
assert JOB.itemsize == 44
SIZES = [(16, 16), (32, 32), (64, 64), (128, 128), (32, 16), (16, 64), (128, 32), (64, 128)]
LAMBDA_MV, NUM_REFP = 1234567, 3  # (lambda_mv = 65536 * sqrt(lambda) of a mid qp; three pictures per list: refi bits 1, 2, 2)
NPIC = 3


def ref_pictures():
    """[refi][list] -> luma plane with PAD samples of margin: refi 0, 1 the textured pictures of the affine MC tests, refi 2 a FLAT picture (zero gradients: the normal
    equations are singular, solve_equal divides 0 by 0)"""
    pics = [[row[l][0] for l in range(2)] for row in A.ref_pictures(1)]
    flat = np.full_like(pics[0][0], 512)
    pics.append([flat, flat.copy()])
    return pics


def org_picture():
    """the picture being coded: the texture of the reference pictures seen through a slight zoom + rotation + shift (so the search has a non-zero affine motion to find), other
    noise"""
    g = np.random.default_rng(4242)
    yy, xx = np.mgrid[0:PIC_H, 0:PIC_W].astype(np.float64)
    xs, ys = PAD + 1.015 * xx + 0.01 * yy + 2.3, PAD + -0.012 * xx + 0.99 * yy - 1.6
    a = 512 + 280 * np.sin(xs / 6.0 + ys / 9.0) + 150 * np.cos(ys / 4.0 - xs / 13.0) + g.integers(-40, 41, size=xx.shape)
    return np.ascontiguousarray(np.clip(a, 0, 1023).astype(np.int16))


def make_jobs(w, h, seed, n=24):
    """n searches of w x h CUs -> (jobs, org_bi [n][h][w] int16: the bi-prediction target of the jobs that have bi set, zeros otherwise)"""
    g = np.random.default_rng(seed)
    jobs = np.zeros(n, JOB)
    org, pics = org_picture(), ref_pictures()
    org_bi = np.zeros((n, h, w), np.int16)
    for i in range(n):
        edge = i % 6
        x = [int(g.integers(0, (PIC_W - w) // 4 + 1)) * 4, 0, PIC_W - w, int(g.integers(0, (PIC_W - w) // 4 + 1)) * 4, PIC_W - w, 0][edge]
        y = [int(g.integers(0, (PIC_H - h) // 4 + 1)) * 4, int(g.integers(0, (PIC_H - h) // 4 + 1)) * 4, 0, PIC_H - h, PIC_H - h, 0][edge]
        j = jobs[i]
        j["x"], j["y"] = x, y
        j["vertex_num"] = 2 + (i // 3) % 2
        j["bi"] = 1 if i % 4 == 3 else 0
        j["refi"] = 2 if i % 11 == 5 else int(g.integers(0, 2))  # (now and then the flat picture)
        j["list"] = int(g.integers(0, 2))
        j["mot_bits_other"] = int(g.integers(0, 40))
        big = [6, 20, 60, 300][(i // 3) % 4]
        spread = [0, 1, 3, 8, 20][(i // 2) % 5] * max(w, h) // 32
        base = g.integers(-big, big + 1, size=2)
        for v in range(3):
            j["mv"][v] = base + (g.integers(-spread, spread + 1, size=2) if v else 0)
            j["mvp"][v] = j["mv"][v] + (g.integers(-3, 4, size=2) if i % 3 else 0)  # (every third job starts AT its predictor: one bit)
        if i % 13 == 7:  # a vector difference beyond the table of bit counts: the exp-Golomb lengths
            j["mvp"][1] = j["mv"][1] + (5000, -2500)
        if j["bi"]:
            other = pics[int(g.integers(0, 2))][1 - int(j["list"])]
            ox, oy = int(g.integers(-3, 4)), int(g.integers(-3, 4))
            org_bi[i] = 2 * org[y:y + h, x:x + w] - other[PAD + y + oy:PAD + y + oy + h, PAD + x + ox:PAD + x + ox + w]
    return jobs, org_bi


def refp_table(pics):
    t = np.zeros(len(pics) * 2, A.REFPIC)
    for r, row in enumerate(pics):
        for l, y in enumerate(row):
            t[r * 2 + l] = (A.plane_ptr(y, 0), 0, 0, 8 * r + l, 0)
    return t


class OracleAffineMe:
    name = "oracle"

    def __init__(self):
        self.L = oracle()
        self.L.xo_affine_me_gradient.restype = None
        self.L.xo_affine_me_gradient.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]

    def run(self, pics, org, jobs, org_bi, w, h, lambda_mv=LAMBDA_MV, num_refp=NUM_REFP):
        """-> (mv [n][3][2], cost [n])"""
        out, t = jobs.copy(), refp_table(pics)
        for i in range(len(out)):
            src = org_bi[i] if out[i]["bi"] else org
            self.L.xo_affine_me_gradient(t.ctypes.data, pics[0][0].shape[1], PIC_W, PIC_H, src.ctypes.data, src.shape[1], out[i:i + 1].ctypes.data, w, h, BD, lambda_mv, num_refp)
        return out["mv"].copy(), out["cost"].copy()


class RefAffineMe:
    name = "reference"

    def __init__(self, simd=1):
        self.L, self.simd = C.CDLL(REF_SO), simd
        self.L.refdrv_affine_me_gradient.restype = C.c_uint

    def run(self, pics, org, jobs, org_bi, w, h, lambda_mv=LAMBDA_MV, num_refp=NUM_REFP):
        n = len(jobs)
        mv, cost = np.zeros((n, 3, 2), np.int16), np.zeros(n, np.uint32)
        for i in range(n):
            j = jobs[i]
            ref = pics[int(j["refi"])][int(j["list"])]
            src = np.ascontiguousarray(org_bi[i]) if j["bi"] else org
            mvp, m = np.ascontiguousarray(j["mvp"]), np.ascontiguousarray(j["mv"]).copy()
            cost[i] = self.L.refdrv_affine_me_gradient(C.c_void_p(A.plane_ptr(ref, 0)), C.c_int(ref.shape[1]), C.c_int(PIC_W), C.c_int(PIC_H), C.c_void_p(src.ctypes.data),
                                                       C.c_int(src.shape[1]), C.c_int(int(j["x"])), C.c_int(int(j["y"])), C.c_int(w.bit_length() - 1), C.c_int(h.bit_length() - 1),
                                                       C.c_int(int(j["refi"])), C.c_int(int(j["list"])), C.c_void_p(mvp.ctypes.data), C.c_void_p(m.ctypes.data), C.c_int(int(j["bi"])),
                                                       C.c_int(int(j["vertex_num"])), C.c_int(BD), C.c_uint(lambda_mv), C.c_int(num_refp), C.c_int(int(j["mot_bits_other"])),
                                                       C.c_int(self.simd))
            mv[i] = m
        return mv, cost


class HostAffineMe:
    """affine_core.h compiled by g++ (tests/native/affine_host.cpp xa_host_affine_me): the kernel's decomposition with its block-wide passes as plain loops"""
    name = "affine_core.h on the host"

    def __init__(self):
        h = A.HostAffine()  # (builds tests/native/build/libaffine_host.so when stale)
        self.H, self.cl = h.H, h.cl
        self.H.xa_host_affine_me.restype = None

    def run(self, pics, org, jobs, org_bi, w, h, lambda_mv=LAMBDA_MV, num_refp=NUM_REFP):
        out, t = jobs.copy(), refp_table(pics)
        self.rounds = np.zeros(len(out), np.int32)
        for i in range(len(out)):
            src = org_bi[i] if out[i]["bi"] else org
            self.H.xa_host_affine_me(C.c_void_p(t.ctypes.data), C.c_int(pics[0][0].shape[1]), C.c_int(PIC_W), C.c_int(PIC_H), C.c_void_p(src.ctypes.data), C.c_int(src.shape[1]),
                                     C.c_void_p(out[i:i + 1].ctypes.data), C.c_int(w), C.c_int(h), C.c_int(BD), C.c_uint32(lambda_mv), C.c_int(num_refp), self.cl,
                                     C.c_void_p(self.rounds[i:i + 1].ctypes.data))
        return out["mv"].copy(), out["cost"].copy()
