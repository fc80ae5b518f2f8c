"""Main profile, affine motion compensation of a CU ("affine MC"): xeve_affine_mc (src_main/xevem_mc.c:2236-2339) = derive_affine_subblock_size_bi, per
list xeve_affine_mc_lc (sub-block interpolation with the Main filters, or the enhanced interpolation filter), the bi-prediction average.
  (cpu) the oracle's restatement against goldens recorded from the reference's own function, and against that function called in place where oracle/_ref exists;
  (cpu) the kernel's per-lane code (xeve_amd/csrc/affine_core.h) compiled for the host against the goldens;
  (gpu) xeve_hip_affine_mc_jobs against oracle and goldens."""
import os

import numpy as np
import pytest

import _affine as A

GOLD = np.load(A.GOLDEN)
PICS = None


def pics():
    global PICS
    if PICS is None:
        PICS = A.ref_pictures(1)
    return PICS


def check(impl, w, h):
    jobs = A.make_jobs(w, h, 7 + w + h)
    Y, U, V, path = impl.run(pics(), jobs, w, h)
    want = GOLD["%dx%d/md5" % (w, h)]
    got = A.digests(Y, U, V)
    bad = [i for i in range(len(jobs)) if not np.array_equal(got[i], want[i])]
    assert not bad, (impl.name, w, h, bad[:8], [tuple(GOLD["%dx%d/path" % (w, h)][i]) for i in bad[:8]])
    if path is not None:
        assert np.array_equal(path, GOLD["%dx%d/path" % (w, h)])
    return Y, U, V


@pytest.mark.parametrize("size", A.SIZES, ids=["%dx%d" % s for s in A.SIZES])
def test_oracle_affine_mc_matches_the_reference_goldens(size):
    check(A.OracleAffine(), *size)


def test_the_cases_take_every_path():
    paths = set()
    for (w, h) in A.SIZES:
        paths |= set((int(p[0]) < 8 or int(p[1]) < 8, int(p[2])) for p in GOLD["%dx%d/path" % (w, h)])
        p = GOLD["%dx%d/path" % (w, h)]
        assert any(int(q[0]) == w and int(q[1]) == h for q in p)  # a CU whose vectors are equal: one block
    # the enhanced filter with and without the range around the centre vector; sub-blocks with the bandwidth condition failed (forced up to 8x8) and passed
    assert paths == {(True, 1), (True, 0), (False, 1), (False, 0)}


@pytest.mark.ref
@pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref/libref_affine.so not built")
@pytest.mark.parametrize("size", [(8, 8), (32, 32), (64, 16), (128, 128)], ids=lambda s: "%dx%d" % s)
def test_oracle_affine_mc_matches_the_reference_in_place(size):
    w, h = size
    jobs = A.make_jobs(w, h, 1000 + w + h, n=32)  # (other seeds than the goldens')
    a, b = A.OracleAffine().run(pics(), jobs, w, h), A.RefAffine().run(pics(), jobs, w, h)
    for k in range(4):
        assert np.array_equal(a[k], b[k]), k
    check(A.RefAffine(), w, h)  # (and the committed goldens are what the reference produces today)


@pytest.mark.parametrize("size", A.SIZES, ids=["%dx%d" % s for s in A.SIZES])
def test_the_kernels_per_lane_code_on_the_host_matches_the_goldens(size):
    """xeve_amd/csrc/affine_core.h (subblock_size, block_vector, mc_sample, eif_range / eif_bilinear / eif_out: what the lanes of affine.hip run) compiled for the host"""
    check(A.HostAffine(), *size)


class HipAffine:
    """xeve_hip_affine_mc_jobs: all jobs of a size in ONE launch, the reference pictures resident in HBM"""
    name = "hip"

    def __init__(self):
        import torch

        import xeve_amd
        from xeve_amd import lib

        xeve_amd.init(0)
        self.t, self.L, self.check, self.dev = torch, lib.load(), lib.check, torch.device("cuda:0")
        self.planes = None

    def run(self, pics_, jobs, w, h):
        t = self.t
        if self.planes is None:  # upload once: [refi][list][c] -> device tensor
            self.planes = [[[t.from_numpy(c).to(self.dev) for c in comps] for comps in row] for row in pics_]
        tab = np.zeros(len(pics_) * 2, A.REFPIC)
        for r, row in enumerate(self.planes):
            for l, comps in enumerate(row):
                ptr = [comps[c].data_ptr() + 2 * ((A.PAD if c == 0 else A.PAD // 2) * (comps[c].shape[1] + 1)) for c in range(3)]
                tab[r * 2 + l] = (ptr[0], ptr[1], ptr[2], 8 * r + l, 0)
        s_l, s_c = A.strides(pics_)
        n = len(jobs)
        d_jobs = t.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(self.dev)
        Y, U, V = (t.full((n * k,), -1, dtype=t.int16, device=self.dev) for k in (h * w, h * w // 4, h * w // 4))
        self.check(self.L.xeve_hip_affine_mc_jobs(tab.ctypes.data, len(pics_), len(pics_), s_l, s_c, A.PIC_W, A.PIC_H, d_jobs.data_ptr(), n, w, h, A.BD, Y.data_ptr(), U.data_ptr(),
                                                  V.data_ptr(), None))
        t.cuda.synchronize()
        return Y.cpu().numpy().reshape(n, h, w), U.cpu().numpy().reshape(n, h // 2, w // 2), V.cpu().numpy().reshape(n, h // 2, w // 2), None


@pytest.fixture(scope="module")
def hip_affine():
    return HipAffine()


@pytest.mark.gpu
@pytest.mark.parametrize("size", A.SIZES, ids=["%dx%d" % s for s in A.SIZES])
def test_hip_affine_mc_matches_oracle_and_goldens(size, hip_affine):
    w, h = size
    Y, U, V = check(hip_affine, w, h)
    a = A.OracleAffine().run(pics(), A.make_jobs(w, h, 7 + w + h), w, h)
    assert np.array_equal(Y, a[0]) and np.array_equal(U, a[1]) and np.array_equal(V, a[2])


class HipAffineHost:
    """xeve_hip_affine_mc_host: ONE CU per call with the reference's arguments and HOST planes (what oracle/ref_shim_affine.c routes the Main encoder's by-name calls of
    xeve_affine_mc to); the table names the pictures the CU uses and nothing else, as the shim's does"""
    name = "hip_host"

    def __init__(self):
        import xeve_amd
        from xeve_amd import lib

        xeve_amd.init(0)
        self.L, self.check = lib.load(), lib.check

    def run(self, pics_, jobs, w, h):
        s_l, s_c = A.strides(pics_)
        n = len(jobs)
        Y, U, V = np.full((n, h, w), -1, np.int16), np.full((n, h // 2, w // 2), -1, np.int16), np.full((n, h // 2, w // 2), -1, np.int16)
        for i, j in enumerate(jobs):
            tab = np.zeros(len(pics_) * 2, A.REFPIC)
            nr = [0, 0]
            for l in range(2):
                r = int(j["refi"][l])
                if r >= 0:
                    comps = pics_[r][l]
                    tab[r * 2 + l] = (A.plane_ptr(comps[0], 0), A.plane_ptr(comps[1], 1), A.plane_ptr(comps[2], 2), 8 * r + l, 0)
                    nr[l] = r + 1
            refi, mv = np.ascontiguousarray(j["refi"]), np.ascontiguousarray(j["mv"])
            self.check(self.L.xeve_hip_affine_mc_host(int(j["x"]), int(j["y"]), A.PIC_W, A.PIC_H, w, h, refi.ctypes.data, mv.ctypes.data, tab.ctypes.data, nr[0], nr[1], s_l, s_c,
                                                      A.PAD, A.PAD // 2, Y[i].ctypes.data, U[i].ctypes.data, V[i].ctypes.data, int(j["vertex_num"]), A.BD))
        return Y, U, V, None


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(8, 8), (64, 64), (16, 8), (32, 128)], ids=["8x8", "64x64", "16x8", "32x128"])
def test_hip_affine_mc_host_form_matches_the_goldens(size):
    """the per-call host form (round 6) over the same jobs: list 0 alone, list 1 alone, both; every path"""
    check(HipAffineHost(), *size)
