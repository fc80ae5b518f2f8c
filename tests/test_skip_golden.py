"""xeve_analyze_skip: the oracle's restatement against the committed reference goldens (runs without the reference)."""
import numpy as np

from _libs import SBAC_DTYPE, SKIP_RESULT_DTYPE, oracle_skip, ptr
from _mc_cases import refpic_table
from _skip_golden import N_CASES, golden


def test_oracle_analyze_skip_matches_reference_goldens():
    O = oracle_skip()
    n, pairs = 0, set()
    for c in golden():
        refs, org = c["refs"], c["org"]
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"],
                             int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
        for i in range(len(c["jobs"])):
            res, best = np.zeros(1, SKIP_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            pr = [np.zeros(c["pred"][k].shape[1], np.int16) for k in range(3)]
            O.xo_analyze_skip(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(c["states"]), c["p"], ptr(c["jobs"][i:i + 1]),
                              ptr(res), ptr(pr[0]), ptr(pr[1]), ptr(pr[2]), ptr(best))
            if c["slice_type"] != 0:
                res["mv"][:, 1] = 0
            assert res.tobytes() == c["res"][i:i + 1].tobytes(), (n, i, res, c["res"][i])
            for k in range(3 if c["idc"] else 1):
                assert np.array_equal(pr[k], c["pred"][k][i]), (n, i, k)
            assert best.tobytes() == c["best"][i:i + 1].tobytes(), (n, i)
            pairs.add((int(res["idx0"][0]), int(res["idx1"][0])))
        n += 1
    assert n == N_CASES and len(pairs) >= 6, pairs
