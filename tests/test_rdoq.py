"""RDOQ: oracle vs the committed reference goldens (CPU), HIP parallel-scan kernel vs goldens and oracle (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from _libs import RdoqEst, oracle_rdoq, ptr
from _rdoq_cases import make_coef, make_est

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdoq_v1.npz")


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        lw, lh, qp, luma, bd, nnz = (int(v) for v in g["p%d" % k])
        est = RdoqEst.from_buffer_copy(np.ascontiguousarray(g["est%d" % k]).tobytes())
        yield lw, lh, qp, luma, bd, float(g["lam%d" % k]), est, np.ascontiguousarray(g["in%d" % k]), g["out%d" % k], nnz


def test_oracle_rdoq_matches_reference_goldens():
    O = oracle_rdoq()
    n = 0
    for lw, lh, qp, luma, bd, lam, est, cin, cout, nnz in golden():
        c = cin.copy()
        assert O.xo_rdoq(ptr(c), lw, lh, qp, lam, luma, bd, 0, C.byref(est)) == nnz
        assert np.array_equal(c, cout), (lw, lh, n)
        n += 1
    assert n == 48


@pytest.mark.gpu
def test_hip_rdoq_matches_reference_goldens():
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    for lw, lh, qp, luma, bd, lam, est, cin, cout, nnz in golden():
        d = torch.from_numpy(cin.reshape(1, -1).copy()).to(dev)
        got_nnz = D.rdoq(d, lw, lh, qp, lam, luma, bd, lib.RdoqEst.from_buffer_copy(bytes(est)))
        assert int(got_nnz[0]) == nnz and np.array_equal(d.cpu().numpy()[0], cout), (lw, lh, qp)


@pytest.mark.gpu
@pytest.mark.parametrize("lw,lh", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 4), (4, 5), (6, 5), (2, 6)])
def test_hip_rdoq_batches_vs_oracle(lw, lh):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O = oracle_rdoq()
    r = np.random.default_rng(1400 + lw * 8 + lh)
    for rep in range(4):
        bd, qp = int(r.choice([8, 10, 12])), int(r.integers(10, 52))
        lam = float(r.choice([0.57, 4.3, 37.1, 220.5, 1500.25])) * (1.0 + float(r.random()))
        luma = int(r.integers(0, 2))
        est = make_est(r)
        blocks = np.stack([make_coef(r, lw, lh, bd, k % 4) for k in range(37)])
        blocks[5] = 0  # sum_all == 0 path
        d = torch.from_numpy(blocks.copy()).to(dev)
        nnz = D.rdoq(d, lw, lh, qp, lam, luma, bd, lib.RdoqEst.from_buffer_copy(bytes(est))).cpu().numpy()
        got = d.cpu().numpy()
        for b in range(len(blocks)):
            e = blocks[b].copy()
            en = O.xo_rdoq(ptr(e), lw, lh, qp, lam, luma, bd, 0, C.byref(est))
            assert nnz[b] == en and np.array_equal(got[b], e), (lw, lh, rep, b, qp, bd)
