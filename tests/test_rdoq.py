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


@pytest.mark.gpu
@pytest.mark.parametrize("lw,lh", [(2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 5)])
def test_residual_chain_with_rdoq_vs_oracle(lw, lh):
    """xeve_hip_residual_rdoq = DIFF, SSD, DCT | zero pre-test + RDOQ | dequant, IDCT, recon, SSD, against the oracle chain"""
    import torch

    import xeve_amd
    from _libs import oracle
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O, OR = oracle(), oracle_rdoq()
    r = np.random.default_rng(1500 + lw * 8 + lh)
    w, h, bd = 1 << lw, 1 << lh, 10
    n, nblk = w * h, 19
    s = 4 * 64 + 32
    org = r.integers(0, 1024, size=(5 * 64 + 16, s), dtype=np.int16)
    pred = r.integers(0, 1024, size=(nblk, n), dtype=np.int16)
    offs = [(8 + (b // 4) * h) * s + 8 + (b % 4) * w for b in range(nblk)]
    for b in range(8):  # good predictions: small residuals, where RDOQ actually decides things
        blk = org.reshape(-1)[np.add.outer(np.arange(h) * s, np.arange(w)).ravel() + offs[b]]
        pred[b] = np.clip(blk + r.integers(-3 * b - 1, 3 * b + 2, size=n), 0, 1023)
    for qp, intra, luma, lam in ((27, 0, 1, 38.5), (37, 1, 0, 310.25), (32, 0, 1, 96.0)):
        est = make_est(r)
        d_org, d_pred = torch.from_numpy(org).to(dev), torch.from_numpy(pred).to(dev)
        jobs = D.make_jobs(offs, np.arange(nblk) * n, dev)
        coef = torch.full((nblk, n), 77, dtype=torch.int16, device=dev)
        rec = torch.zeros_like(d_org)
        nnz = torch.zeros(nblk, dtype=torch.int32, device=dev)
        ssd = torch.zeros((nblk, 2), dtype=torch.int64, device=dev)
        D.residual_rdoq(d_org, s, d_pred, w, jobs, lw, lh, bd, qp, intra, lam, luma, lib.RdoqEst.from_buffer_copy(bytes(est)), coef, rec, s, nnz, ssd)
        coef, rec, nnz, ssd = coef.cpu().numpy(), rec.cpu().numpy(), nnz.cpu().numpy(), ssd.cpu().numpy()
        qs, dqs = D.QUANT_SCALE[0][qp % 6], D.DQ_SCALE[qp % 6] << (qp // 6)
        coded = 0
        for b in range(nblk):
            c = np.zeros(n, np.int16)
            O.xo_diff(w, h, ptr(org, offs[b]), ptr(pred, b * n), s, w, w, ptr(c))
            assert ssd[b, 0] == O.xo_ssd(w, h, ptr(org, offs[b]), ptr(pred, b * n), s, w, bd)
            O.xo_trans(ptr(c), lw, lh, bd)
            if O.xo_rdoq_zero_test(ptr(c), lw, lh, qp, qs, intra, bd):
                e_nnz = OR.xo_rdoq(ptr(c), lw, lh, qp, lam, luma, bd, 0, C.byref(est))
            else:
                c[:] = 0
                e_nnz = 0
            assert nnz[b] == e_nnz and np.array_equal(coef[b], c), ("levels", lw, lh, qp, b)
            coded += e_nnz
            O.xo_dquant(ptr(c), lw, lh, dqs, bd)
            O.xo_itrans(ptr(c), lw, lh, bd)
            e = np.zeros((h, s), np.int16)
            O.xo_recon(ptr(c), ptr(pred, b * n), 1, w, h, s, ptr(e), bd)
            y0, x0 = offs[b] // s, offs[b] % s
            assert np.array_equal(rec[y0:y0 + h, x0:x0 + w], e[:, :w]), ("rec", lw, lh, qp, b)
            assert ssd[b, 1] == O.xo_ssd(w, h, ptr(org, offs[b]), ptr(rec, offs[b]), s, s, bd)
        assert coded > 0


@pytest.mark.gpu
def test_hip_rdoq_bit_est_vs_goldens_and_oracle():
    """xeve_rdoq_bit_est on the device (entropy table built by the library) == the reference's (goldens) == the oracle's"""
    import torch

    import xeve_amd
    from _libs import EST_FULL_INTS, SBAC_DTYPE, oracle_sbac
    from _sbac_cases import make_states
    from _sbac_golden import GOLD, est_states
    from xeve_amd import device as D

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    st = est_states().view(np.uint8)
    got = D.rdoq_bit_est(torch.from_numpy(st.copy()).to(dev)).cpu().numpy()
    assert np.array_equal(got, g["est"])
    O = oracle_sbac()
    states = make_states(np.random.default_rng(5), 300)
    got = D.rdoq_bit_est(torch.from_numpy(states.view(np.uint8).copy()).to(dev)).cpu().numpy()
    for i in range(len(states)):
        e = np.zeros(EST_FULL_INTS, np.int32)
        O.xo_rdoq_bit_est(ptr(states[i:i + 1]), ptr(e))
        assert np.array_equal(got[i], e), i


@pytest.mark.gpu
@pytest.mark.parametrize("lw,lh", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (2, 4), (5, 3)])
def test_hip_rdoq_with_device_estimates_vs_oracle(lw, lh):
    """the loop the reference closes per CU: coder state -> xeve_rdoq_bit_est -> RDOQ, estimates picked per block on the device"""
    import torch

    import xeve_amd
    from _libs import EST_FULL_INTS, oracle_sbac
    from _sbac_cases import make_states
    from xeve_amd import device as D

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O, OS = oracle_rdoq(), oracle_sbac()
    r = np.random.default_rng(3300 + lw * 8 + lh)
    states = make_states(r, 9)
    est_dev = D.rdoq_bit_est(torch.from_numpy(states.view(np.uint8).copy()).to(dev))
    for ch_type in range(3):
        for is_intra in (0, 1):
            slice_i = int(r.integers(0, 2))  # the zero pre-test follows the slice type, the cbf pair the CU's own mode
            bd, qp = int(r.choice([8, 10])), int(r.integers(14, 48))
            lam = float(r.choice([0.9, 11.3, 140.5])) * (1.0 + float(r.random()))
            blocks = np.stack([make_coef(r, lw, lh, bd, k % 4) for k in range(41)])
            idx = r.integers(0, len(states), size=len(blocks)).astype(np.int32)
            d = torch.from_numpy(blocks.copy()).to(dev)
            nnz = D.rdoq_dev(d, lw, lh, qp, lam, ch_type, bd, est_dev, torch.from_numpy(idx).to(dev), zero_test=True, is_intra_slice=bool(slice_i),
                             is_intra_cu=bool(is_intra)).cpu().numpy()
            got = d.cpu().numpy()
            qs = D.QUANT_SCALE[0][qp % 6]
            for b in range(len(blocks)):
                full = np.zeros(EST_FULL_INTS, np.int32)
                OS.xo_rdoq_bit_est(ptr(states[idx[b]:idx[b] + 1]), ptr(full))
                est = RdoqEst()
                OS.xo_rdoq_est_select(ptr(full), ch_type, is_intra, C.byref(est))
                e = blocks[b].copy()
                if O.xo_rdoq_zero_test(ptr(e), lw, lh, qp, qs, slice_i, bd):
                    en = O.xo_rdoq(ptr(e), lw, lh, qp, lam, int(ch_type == 0), bd, 0, C.byref(est))
                else:
                    e[:], en = 0, 0
                assert nnz[b] == en and np.array_equal(got[b], e), (lw, lh, ch_type, is_intra, b)
            # record 0 for every block when no index is given
            d2 = torch.from_numpy(blocks.copy()).to(dev)
            nnz2 = D.rdoq_dev(d2, lw, lh, qp, lam, ch_type, bd, est_dev, None, zero_test=False, is_intra_cu=bool(is_intra)).cpu().numpy()
            full = np.zeros(EST_FULL_INTS, np.int32)
            OS.xo_rdoq_bit_est(ptr(states[0:1]), ptr(full))
            est = RdoqEst()
            OS.xo_rdoq_est_select(ptr(full), ch_type, is_intra, C.byref(est))
            for b in range(0, len(blocks), 5):
                e = blocks[b].copy()
                assert nnz2[b] == O.xo_rdoq(ptr(e), lw, lh, qp, lam, int(ch_type == 0), bd, 0, C.byref(est)) and np.array_equal(d2[b].cpu().numpy(), e)
