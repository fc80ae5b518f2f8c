"""Pins xo_pinter_analyze_cu against the reference's xeve_pinter_analyze_cu (xeve_pinter.c:1839-2047, = ctx->fn_pinter_analyze_cu) run through
oracle/_ref/libref_rdo.so: returned cost (bit pattern), cu_mode, motion data, core->nnz, coefficients, reconstruction and core->s_next_best, with
the reference's own candidate derivation (xeve_get_motion, xeve_get_mv_dir), motion search (pinter_me_epzs), check_best_mvp, analyze_bi."""
import numpy as np
import pytest

from _inter_cases import make_inter_jobs, make_inter_params, make_inter_picture, mask_unobservable
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, oracle_inter, ptr, ref_inter
from _mc_cases import refpic_table
from _rdo_cases import states

pytestmark = pytest.mark.skipif(ref_inter() is None, reason="oracle/_ref not built (no /root/reference here)")


def run_both(w, h, bd, nref, idc, slice_type, skip_th, sizes, n, seed, nref1=None):
    O, R = oracle_inter(), ref_inter()
    r = np.random.default_rng(seed)
    refs, org = make_inter_picture(r, w, h, bd, nref, idc, slice_type)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 5)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                        np.uint64)
    modes = []
    for lw in sizes:
        cu = 1 << lw
        P = make_inter_params(r, lw, w, h, bd, nref, idc, slice_type, refs, skip_th, nref1=nref1)
        jobs = make_inter_jobs(r, n, w, h, cu, len(st), refs, slice_type)
        nc = max(1, (cu >> refs["ws"]) * (cu >> refs["hs"]))
        for i in range(len(jobs)):
            ra, rb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, INTER_RESULT_DTYPE)
            ca = [np.zeros(cu * cu, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            cb = [x.copy() for x in ca]
            pa = [x.copy() for x in ca]
            pb = [x.copy() for x in ca]
            ba, bb = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(ra),
                                   ptr(ca[0]), ptr(ca[1]), ptr(ca[2]), ptr(pa[0]), ptr(pa[1]), ptr(pa[2]), ptr(ba))
            R.refdrv_pinter_analyze_cu(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab),
                                       refs["s_l"], refs["s_c"], ptr(st), P, refs["gop"], ptr(jobs[i:i + 1]), ptr(rb), ptr(cb[0]), ptr(cb[1]), ptr(cb[2]),
                                       ptr(pb[0]), ptr(pb[1]), ptr(pb[2]), ptr(bb))
            key = (lw, i, jobs[i], ra[0], rb[0])
            ma, mb = mask_unobservable(ra, slice_type), mask_unobservable(rb, slice_type)
            assert ma.tobytes() == mb.tobytes(), key
            skip = int(ra["cu_mode"][0]) == 2
            for k in range(3 if idc else 1):
                assert skip or np.array_equal(ca[k], cb[k]), (k,) + key
                assert np.array_equal(pa[k], pb[k]), (k,) + key
            assert ba.tobytes() == bb.tobytes(), key
            modes.append(int(ra["best_idx"][0]))
    return modes


@pytest.mark.parametrize("w,h,bd,nref,idc,slice_type", [(128, 96, 10, 2, 1, 0), (128, 64, 10, 2, 1, 1), (96, 64, 8, 1, 1, 0), (64, 64, 10, 2, 0, 0),
                                                        (192, 128, 10, 3, 1, 0), (128, 128, 10, 4, 1, 1), (96, 64, 10, 2, 3, 0)])
def test_pinter_analyze_cu(w, h, bd, nref, idc, slice_type):
    modes = run_both(w, h, bd, nref, idc, slice_type, 0.0, [3, 4, 5, 6], 25, 11 * w + h + bd + nref + idc + slice_type)
    assert len(set(modes)) >= (3 if slice_type == 0 else 2), modes


def test_pinter_analyze_cu_skip_threshold():
    """a positive skip_th ends the analysis after the skip mode for CUs whose skip residual is small"""
    modes = run_both(128, 96, 10, 2, 1, 0, 6.0, [3, 4, 5], 12, 5)
    assert 3 in modes and len(set(modes)) >= 2, modes


def test_pinter_analyze_cu_same_with_the_simd_tables():
    """the driver can let the reference pick its SSE / AVX2 tables (the CPU baseline is timed that way): same results as with the plain-C tables"""
    R = ref_inter()
    R.refdrv_set_simd(1)
    try:
        modes = run_both(128, 96, 10, 2, 1, 0, 0.0, [3, 4, 5, 6], 8, 321)
    finally:
        R.refdrv_set_simd(0)
    assert len(set(modes)) >= 2


def test_pinter_analyze_cu_list1_shorter_than_list0():
    """three pictures in list 0, two in list 1: the uni-directional searches walk each list's own count, analyze_bi walks BOTH lists with list 1's
    (pi->num_refp as the list-1 search left it) and prices the reference index with list 1's table"""
    modes = run_both(128, 96, 10, 3, 1, 0, 0.0, [3, 4, 5], 20, 780, nref1=2)
    assert 2 in modes, modes


def test_pinter_analyze_cu_fuzz_vs_reference():
    """the oracle against the reference over random configurations incl. 12-bit, 4:4:4, QP / lambda extremes, candidates far outside the picture
    with the CU on a picture corner, all candidates equal, list 1 shorter than list 0, positive skip_th (_inter_cases.fuzz_cases)"""
    from _inter_cases import fuzz_cases

    O, R = oracle_inter(), ref_inter()
    total, kinds = 0, set()
    for refs, org, st, P, jobs, meta in fuzz_cases(16, 500, n_jobs=14):
        cu, idc = 1 << meta["lw"], meta["idc"]
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                            np.uint64)
        nc = max(1, (cu >> refs["ws"]) * (cu >> refs["hs"]))
        for i in range(len(jobs)):
            ra, rb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, INTER_RESULT_DTYPE)
            ca = [np.zeros(cu * cu, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            cb, pa, pb = [x.copy() for x in ca], [x.copy() for x in ca], [x.copy() for x in ca]
            ba, bb = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(ra), ptr(ca[0]),
                                   ptr(ca[1]), ptr(ca[2]), ptr(pa[0]), ptr(pa[1]), ptr(pa[2]), ptr(ba))
            R.refdrv_pinter_analyze_cu(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab),
                                       refs["s_l"], refs["s_c"], ptr(st), P, refs["gop"], ptr(jobs[i:i + 1]), ptr(rb), ptr(cb[0]), ptr(cb[1]), ptr(cb[2]), ptr(pb[0]),
                                       ptr(pb[1]), ptr(pb[2]), ptr(bb))
            key = (meta, i, jobs[i], ra[0], rb[0])
            assert mask_unobservable(ra, meta["slice_type"]).tobytes() == mask_unobservable(rb, meta["slice_type"]).tobytes(), key
            skip = int(ra["cu_mode"][0]) == 2
            for k in range(3 if idc else 1):
                assert (skip or np.array_equal(ca[k], cb[k])) and np.array_equal(pa[k], pb[k]), (k,) + key
            assert ba.tobytes() == bb.tobytes(), key
            total += 1
        kinds.add(meta["kind"])
    assert total > 800 and kinds == {0, 1, 2, 3}
