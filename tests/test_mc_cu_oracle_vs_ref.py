"""Pins xo_mc_cu against the reference's exported xeve_mc (through oracle/_ref/libref_df.so, which only adapts the structs)."""
import numpy as np
import pytest

from _libs import oracle_mc_cu, ptr, ref_mc_cu
from _mc_cases import make_jobs, make_refs, refpic_table

pytestmark = pytest.mark.skipif(ref_mc_cu() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("w,h,bd,idc,nref", [(128, 96, 10, 1, 2), (64, 64, 8, 1, 1), (96, 64, 10, 3, 2), (72, 48, 10, 0, 3), (128, 64, 12, 1, 2)])
def test_mc_cu(w, h, bd, idc, nref):
    O, R = oracle_mc_cu(), ref_mc_cu()
    r = np.random.default_rng(w + h + bd + idc)
    refs = make_refs(r, w, h, bd, nref, idc)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    for (cuw, cuh) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (4, 4)]:
        if cuw > w or cuh > h:
            continue
        jobs = make_jobs(r, 40, w, h, cuw, cuh, nref)
        cw, ch = cuw >> refs["ws"], cuh >> refs["hs"]
        for i in range(len(jobs)):
            a = [np.full(cuw * cuh, -1, np.int16), np.full(cw * ch, -1, np.int16), np.full(cw * ch, -1, np.int16)]
            b = [x.copy() for x in a]
            O.xo_mc_cu(ptr(tab), refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), cuw, cuh, bd, bd, idc, ptr(a[0]), ptr(a[1]), ptr(a[2]))
            R.refdrv_mc_cu(ptr(tab), nref, refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), cuw, cuh, bd, bd, idc, ptr(b[0]), ptr(b[1]), ptr(b[2]))
            for k in range(3 if idc else 1):
                assert np.array_equal(a[k], b[k]), (cuw, cuh, i, k, jobs[i])
