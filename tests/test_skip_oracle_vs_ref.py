"""Pins xo_analyze_skip against the reference's static xeve_analyze_skip (xeve_pinter.c:1337-1530) run through oracle/_ref/libref_rdo.so, the
merge candidates derived by the reference's own xeve_get_motion from neighbour maps loaded with the test's vectors: returned cost (bit pattern),
winning candidate pair, vectors, best SSD, the kept prediction and core->s_temp_best."""
import numpy as np
import pytest

from _libs import SBAC_DTYPE, SKIP_RESULT_DTYPE, oracle_skip, ptr, ref_skip
from _mc_cases import refpic_table
from _rdo_cases import make_params, make_picture, make_skip_jobs, states

pytestmark = pytest.mark.skipif(ref_skip() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("w,h,bd,nref,idc,slice_type,ncand", [(128, 96, 10, 2, 1, 0, 3), (128, 64, 10, 1, 1, 1, 2), (96, 64, 8, 2, 1, 0, 4), (64, 64, 10, 1, 0, 0, 3)])
def test_analyze_skip(w, h, bd, nref, idc, slice_type, ncand):
    O, R = oracle_skip(), ref_skip()
    r = np.random.default_rng(w + h + bd + nref + idc + slice_type + ncand)
    refs, org = make_picture(r, w, h, bd, nref, idc)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 5)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                        np.uint64)
    pairs = set()
    for (lw, lh) in [(3, 3), (4, 4), (5, 5), (6, 6)]:
        cuw, cuh = 1 << lw, 1 << lh
        if cuw > w or cuh > h:
            continue
        p = make_params(r, lw, lh, w, h, bd, nref, idc, slice_type)
        jobs = make_skip_jobs(r, 30, w, h, cuw, cuh, len(st), ncand)
        nc = (cuw >> refs["ws"]) * (cuh >> refs["hs"])
        for i in range(len(jobs)):
            ra, rb = np.zeros(1, SKIP_RESULT_DTYPE), np.zeros(1, SKIP_RESULT_DTYPE)
            pa = [np.zeros(cuw * cuh, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            pb = [x.copy() for x in pa]
            ba, bb = np.zeros(1, SBAC_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_analyze_skip(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(ra), ptr(pa[0]),
                              ptr(pa[1]), ptr(pa[2]), ptr(ba))
            R.refdrv_analyze_skip(ptr(org[0], refs["org_l"]), ptr(org[1], refs["org_c"]), ptr(org[2], refs["org_c"]), refs["s_l"], refs["s_c"], ptr(tab),
                                  refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(rb), ptr(pb[0]), ptr(pb[1]), ptr(pb[2]), ptr(bb))
            key = (lw, i, jobs[i], ra, rb)
            if slice_type != 0:  # P slice: list 1 is never derived, pi->mv[PRED_SKIP][REFP_1] holds whatever pi->mvp[REFP_1] held before
                ra["mv"][0, 1], rb["mv"][0, 1] = 0, 0
            assert ra.tobytes() == rb.tobytes(), key
            for k in range(3 if idc else 1):
                assert np.array_equal(pa[k], pb[k]), (k,) + key
            assert ba.tobytes() == bb.tobytes(), key
            pairs.add((int(ra["idx0"][0]), int(ra["idx1"][0])))
    assert len(pairs) >= (3 if slice_type == 0 else 2), pairs
