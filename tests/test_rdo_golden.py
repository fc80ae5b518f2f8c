"""pinter_residue_rdo: the oracle's composition against the committed reference goldens (runs without the reference)."""
import numpy as np

from _libs import RDO_RESULT_DTYPE, SBAC_DTYPE, oracle_rdo, ptr
from _mc_cases import refpic_table
from _rdo_golden import golden


def test_oracle_residue_rdo_matches_reference_goldens():
    O = oracle_rdo()
    n = 0
    for c in golden():
        refs, org = c["refs"], c["org"]
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"],
                             int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
        for i in range(len(c["jobs"])):
            res, best = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            co = [np.zeros(c["coef"][k].shape[1], np.int16) for k in range(3)]
            O.xo_residue_rdo(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(c["states"]), c["p"], ptr(c["jobs"][i:i + 1]),
                             ptr(res), ptr(co[0]), ptr(co[1]), ptr(co[2]), ptr(best))
            assert res["cost"][0].tobytes() == c["cost"][i].tobytes() and np.array_equal(res["nnz"][0], c["nnz"][i]), (n, i)
            for k in range(3 if c["idc"] else 1):
                assert np.array_equal(co[k], c["coef"][k][i]), (n, i, k)
            assert best.tobytes() == c["best"][i:i + 1].tobytes(), (n, i)
        n += 1
    assert n == 6
