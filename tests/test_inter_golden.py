"""xeve_pinter_analyze_cu: the oracle's restatement against the committed reference goldens (runs without the reference)."""
import numpy as np

from _inter_cases import mask_unobservable
from _inter_golden import CASES, golden
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, oracle_inter, ptr
from _mc_cases import refpic_table


def test_oracle_pinter_analyze_cu_matches_reference_goldens():
    O = oracle_inter()
    n, modes = 0, set()
    for c in golden():
        refs, org = c["refs"], c["org"]
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"],
                             int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
        for i in range(len(c["jobs"])):
            res, best = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            co = [np.zeros(c["coef"][k].shape[1], np.int16) for k in range(3)]
            rc = [np.zeros(c["coef"][k].shape[1], np.int16) for k in range(3)]
            O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(c["states"]), c["P"], ptr(c["jobs"][i:i + 1]),
                                   ptr(res), ptr(co[0]), ptr(co[1]), ptr(co[2]), ptr(rc[0]), ptr(rc[1]), ptr(rc[2]), ptr(best))
            modes.add(int(res["best_idx"][0]))
            assert mask_unobservable(res, c["slice_type"]).tobytes() == c["res"][i:i + 1].tobytes(), (n, i, res, c["res"][i])
            for k in range(3 if c["idc"] else 1):
                assert np.array_equal(co[k], c["coef"][k][i]) and np.array_equal(rc[k], c["rec"][k][i]), (n, i, k)
            assert best.tobytes() == c["best"][i:i + 1].tobytes(), (n, i)
        n += 1
    assert n == len(CASES) and modes == {0, 1, 2, 3, 4}, modes
