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
