"""Pins xo_residue_rdo -- the composition of the oracle's prediction, residual, transform, RDOQ, reconstruction and CABAC bit
counting with the coded-block-flag decision -- against the reference's static pinter_residue_rdo (xeve_pinter.c:906-1336) run on
the same pictures and coder state through oracle/_ref/libref_rdo.so: returned cost (bit pattern of the double), core->nnz,
the coefficient buffers and core->s_temp_best."""
import numpy as np
import pytest

from _libs import RDO_RESULT_DTYPE, SBAC_DTYPE, oracle_rdo, ptr, ref_rdo
from _mc_cases import refpic_table
from _rdo_cases import make_jobs, make_params, make_picture, states

pytestmark = pytest.mark.skipif(ref_rdo() is None, reason="oracle/_ref not built (no /root/reference here)")


def run_both(r, w, h, bd, nref, idc, slice_type, sizes, njobs):
    O, R = oracle_rdo(), ref_rdo()
    refs, org = make_picture(r, w, h, bd, nref, idc)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 6)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"],
                         int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    stats = dict(n=0, zero=0, dropped=0, kept=0)
    for (lw, lh) in sizes:
        cuw, cuh = 1 << lw, 1 << lh
        if cuw > w or cuh > h:
            continue
        p = make_params(r, lw, lh, w, h, bd, nref, idc, slice_type)
        jobs = make_jobs(r, njobs, w, h, cuw, cuh, nref, len(st), slice_type)
        ws, hs = refs["ws"], refs["hs"]
        nc = (cuw >> ws) * (cuh >> hs)
        for i in range(len(jobs)):
            ra, rb = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, RDO_RESULT_DTYPE)
            ca = [np.zeros(cuw * cuh, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            cb = [x.copy() for x in ca]
            ba, bb = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_residue_rdo(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(ra),
                             ptr(ca[0]), ptr(ca[1]), ptr(ca[2]), ptr(ba))
            R.refdrv_residue_rdo(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab),
                                 refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(rb), ptr(cb[0]), ptr(cb[1]), ptr(cb[2]), ptr(bb))
            key = (lw, lh, i, jobs[i], ra, rb)
            assert ra["cost"].tobytes() == rb["cost"].tobytes(), key
            assert np.array_equal(ra["nnz"], rb["nnz"]), key
            for k in range(3 if idc else 1):
                assert np.array_equal(ca[k], cb[k]), (k,) + key
            assert ba.tobytes() == bb.tobytes(), key
            stats["n"] += 1
            stats["zero"] += int(not ra["nnz"].any())
            nz = ra["nnz"][0][:3 if idc else 1]
            stats["kept"] += int(nz.all())
            stats["dropped"] += int(nz.any() and not nz.all())
    return stats


@pytest.mark.parametrize("w,h,bd,nref,idc,slice_type", [(128, 96, 10, 2, 1, 0), (128, 64, 10, 1, 1, 1), (96, 64, 8, 2, 1, 0), (64, 64, 10, 2, 0, 0),
                                                        (128, 128, 10, 3, 1, 0)])
def test_residue_rdo(w, h, bd, nref, idc, slice_type):
    r = np.random.default_rng(w + h + bd + nref + idc + slice_type)
    tot = dict(n=0, zero=0, dropped=0, kept=0)
    for rep in range(2):
        s = run_both(r, w, h, bd, nref, idc, slice_type, [(3, 3), (4, 4), (5, 5), (6, 6), (2, 2)], 20)
        for k in tot:
            tot[k] += s[k]
    assert tot["n"] >= 100 and tot["zero"] > 0 and tot["kept"] > 0, tot  # the cases reach the different branches of the decision


def test_residue_rdo_fuzz_vs_reference():
    """pinter_residue_rdo, oracle against the reference, over _rdo_cases.fuzz_cases: 12-bit, 4:4:4, QP 0 ... 51 + 6 (bd - 8), lambda 1e-3 ... 5e4, vectors far
    outside the picture with the CU on a picture corner, non-square CUs"""
    from _rdo_cases import fuzz_cases

    O, R = oracle_rdo(), ref_rdo()
    total, kinds = 0, set()
    for refs, org, st, p, lw, lh, jobs, sj, ncand, meta in fuzz_cases(10, 300, n_jobs=16):
        cuw, cuh, idc = 1 << lw, 1 << lh, meta["idc"]
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                            np.uint64)
        nc = max(1, (cuw >> refs["ws"]) * (cuh >> refs["hs"]))
        for i in range(len(jobs)):
            ra, rb = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, RDO_RESULT_DTYPE)
            ca = [np.zeros(cuw * cuh, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            cb = [x.copy() for x in ca]
            ba, bb = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_residue_rdo(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(ra), ptr(ca[0]), ptr(ca[1]),
                             ptr(ca[2]), ptr(ba))
            R.refdrv_residue_rdo(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"],
                                 refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(rb), ptr(cb[0]), ptr(cb[1]), ptr(cb[2]), ptr(bb))
            key = (meta, i, jobs[i], ra[0], rb[0])
            assert ra["cost"].tobytes() == rb["cost"].tobytes() and np.array_equal(ra["nnz"], rb["nnz"]), key
            for k in range(3 if idc else 1):
                assert np.array_equal(ca[k], cb[k]), (k,) + key
            assert ba.tobytes() == bb.tobytes(), key
            total += 1
        kinds.add(meta["kind"])
    assert total > 800 and kinds == {0, 1, 2}
