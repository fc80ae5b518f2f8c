"""Pins the oracle's CABAC restatement (xo_sbac_*, xo_eco_run_length_cc, xo_cu_bits) against the UNMODIFIED reference
(oracle/_ref/libref_sbac.so wraps xeve_sbac_encode_bin, xeve_eco_run_length_cc, xeve_rdo_bit_cnt_cu_inter/_comp/_skip)."""
import numpy as np
import pytest

from _libs import SBAC_DTYPE, oracle_sbac, ptr, ref_sbac
from _sbac_cases import clamp_refi, make_jobs, make_levels, make_params, make_states

pytestmark = pytest.mark.skipif(ref_sbac() is None, reason="oracle/_ref not built (no /root/reference here)")


def same(a, b):
    for f in SBAC_DTYPE.names:
        assert np.array_equal(a[f], b[f]), f


def test_bins_random_walk():
    """long random bin sequences: every field of the coder state after each bin"""
    O, R = oracle_sbac(), ref_sbac()
    r = np.random.default_rng(11)
    for trial in range(6):
        a = make_states(r, 2)[1:2].copy()
        a["code_bits"], a["stacked_ff"], a["bitcounter"] = 11, 0, 0
        b = a.copy()
        for k in range(4000):
            ci, ep = int(r.integers(0, 68)), int(r.random() < 0.25)
            bias = 0.08 if trial % 2 else 0.5  # skewed sources drive the models into their corners
            bit = int(r.random() < bias)
            if ep:
                O.xo_sbac_bin_ep(ptr(a), bit)
            else:
                O.xo_sbac_bin(ptr(a), ci, bit)
            R.refdrv_sbac_bin(ptr(b), ci, bit, ep)
            same(a, b)
        assert O.xo_sbac_bits(ptr(a)) > 0


@pytest.mark.parametrize("cm_init", [0, 1])
def test_run_length_cc(cm_init):
    O, R = oracle_sbac(), ref_sbac()
    r = np.random.default_rng(12 + cm_init)
    for lw in range(1, 7):
        for lh in range(1, 7):
            for kind in range(5):
                c = make_levels(r, 1 << (lw + lh), kind)
                nnz = int(np.count_nonzero(c))
                for num_sig in {nnz, max(1, nnz - 1), nnz + 3}:
                    a = make_states(r, 2)[1:2].copy()
                    O.xo_sbac_bit_reset(ptr(a))
                    b = a.copy()
                    ch = int(r.integers(0, 2))
                    O.xo_eco_run_length_cc(ptr(a), ptr(c), lw, lh, num_sig, ch, cm_init)
                    R.refdrv_run_length_cc(ptr(b), ptr(c), lw, lh, num_sig, ch, cm_init)
                    same(a, b)


@pytest.mark.parametrize("slice_type,num_refp,cm_init,idc", [(0, (2, 2), 0, 1), (1, (1, 0), 0, 1), (0, (4, 3), 0, 1), (0, (3, 1), 1, 1),
                                                             (2, (0, 0), 0, 1), (0, (2, 2), 0, 0), (0, (2, 2), 0, 2), (0, (2, 2), 0, 3)])
def test_cu_bits(slice_type, num_refp, cm_init, idc):
    O, R = oracle_sbac(), ref_sbac()
    r = np.random.default_rng(13 + slice_type + 10 * cm_init + 100 * idc)
    states = make_states(r, 8)
    for lw in range(2, 7):
        for lh in range(2, 7):
            p = make_params(lw, lh, slice_type, num_refp, cm_init, idc)
            jobs, coef = make_jobs(r, 10, lw, lh, len(states), idc, nnz_mode=(lw + lh) & 1)
            clamp_refi(jobs, num_refp)
            for i in range(len(jobs)):
                a, b = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
                ba = O.xo_cu_bits(ptr(states), ptr(a), p, ptr(jobs[i:i + 1]), ptr(coef))
                bb = R.refdrv_cu_bits(ptr(states), ptr(b), p, ptr(jobs[i:i + 1]), ptr(coef))
                assert ba == bb, (lw, lh, i, jobs[i])
                same(a, b)


def test_rdoq_bit_est_and_entropy_table():
    """xo_rdoq_bit_est == the reference's static xeve_rdoq_bit_est on the same coder state; entropy_bits table identical"""
    from _libs import EST_FULL_INTS

    O, R = oracle_sbac(), ref_sbac()
    tab = np.zeros(1024, np.int32)
    R.refdrv_entropy_bits(ptr(tab))
    assert [O.xo_entropy_bits(i) for i in range(1024)] == tab.tolist()
    r = np.random.default_rng(21)
    st = make_states(r, 200)
    for i in range(len(st)):
        a, b = np.zeros(EST_FULL_INTS, np.int32), np.zeros(EST_FULL_INTS, np.int32)
        O.xo_rdoq_bit_est(ptr(st[i:i + 1]), ptr(a))
        R.refdrv_rdoq_bit_est(ptr(st[i:i + 1]), ptr(b))
        assert np.array_equal(a, b), i


@pytest.mark.parametrize("idc", [1, 0])
def test_eco_coef_alone(idc):
    """job mode 5 = ctx->fn_eco_coef (xeve_eco_coef) on its own: inter / intra cbf syntax, any subset of components, b_no_cbf, and
    continuing the coder where the state stands (no xeve_sbac_bit_reset)"""
    O, R = oracle_sbac(), ref_sbac()
    r = np.random.default_rng(31 + idc)
    for lw in range(2, 7):
        for lh in range(2, 7):
            p = make_params(lw, lh, 0, (2, 2), 0, idc)
            jobs, coef = make_jobs(r, 24, lw, lh, 8, idc)
            jobs["mode"] = 5
            states = make_states(r, 8)
            for i in range(len(jobs)):
                runs = int(r.integers(1, 8)) if idc else 1
                intra, nocbf, noreset = int(r.random() < 0.4), int(r.random() < 0.2), int(r.random() < 0.5)
                if nocbf and not (intra or any(jobs["nnz"][i][c] for c in range(3) if (runs >> c) & 1)):
                    nocbf = 0  # the reference asserts cbf_all != 0 when the flag is implied
                jobs["dir_flag"][i] = intra | (nocbf << 1) | (runs << 2) | (noreset << 5)
                if noreset:  # a state in mid-stream, as it stands after earlier bins
                    states["code_bits"] = r.integers(1, 9, size=len(states))
                    states["code"] = r.integers(0, 1 << 17, size=len(states)) << (8 - states["code_bits"]).astype(np.uint32)
                    states["is_pending_byte"], states["pending_byte"] = 1, r.integers(0, 256, size=len(states))
                    states["stacked_ff"], states["stacked_zero"] = r.integers(0, 3, size=len(states)), r.integers(0, 3, size=len(states))
                a, b = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
                ba = O.xo_cu_bits(ptr(states), ptr(a), p, ptr(jobs[i:i + 1]), ptr(coef))
                bb = R.refdrv_cu_bits(ptr(states), ptr(b), p, ptr(jobs[i:i + 1]), ptr(coef))
                assert ba == bb, (lw, lh, i, jobs[i])
                same(a, b)


@pytest.mark.parametrize("slice_type,idc", [(2, 1), (0, 1), (1, 1), (2, 0), (0, 3)])
def test_cu_bits_intra_syntax(slice_type, idc):
    """job modes 7 / 8 / 9 = xeve_rdo_bit_cnt_cu_intra / _cu_intra_luma / _intra_dir (xeve_mode.c:81-175): skip flag and pred_mode outside I slices, the
    prediction mode as the unary index mpm[ipm] over the two intra_dir models, intra cbf flags, coefficients"""
    O, R = oracle_sbac(), ref_sbac()
    r = np.random.default_rng(41 + slice_type + 10 * idc)
    states = make_states(r, 8)
    for lw in range(2, 7):
        p = make_params(lw, lw, slice_type, (2, 2), 0, idc)
        jobs, coef = make_jobs(r, 30, lw, lw, len(states), idc)
        jobs["mode"] = r.integers(7, 10, size=len(jobs))
        jobs["mvp_idx"][:, 0] = r.integers(0, 5, size=len(jobs))
        jobs["nnz"][jobs["mode"] == 8, 1:] = 0  # luma alone: the chroma counts are cleared (xeve_sub_block_tq ran for Y only)
        for i in range(len(jobs)):
            a, b = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            ba = O.xo_cu_bits(ptr(states), ptr(a), p, ptr(jobs[i:i + 1]), ptr(coef))
            bb = R.refdrv_cu_bits(ptr(states), ptr(b), p, ptr(jobs[i:i + 1]), ptr(coef))
            assert ba == bb, (lw, i, jobs[i])
            same(a, b)
