"""a8 (xeve_mc): the oracle against the committed reference goldens (reference pictures regenerated from the recorded seed)."""
import os

import numpy as np

from _libs import CU_MC_JOB_DTYPE, oracle_mc_cu, ptr
from _mc_cases import make_jobs, make_refs, refpic_table

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_cu_v1.npz")


def test_oracle_mc_cu_matches_reference_goldens():
    O = oracle_mc_cu()
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        seed, w, h, bd, idc, nref, cuw, cuh = (int(v) for v in g["p%d" % k])
        r = np.random.default_rng(seed)
        refs = make_refs(r, w, h, bd, nref, idc)
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        jobs = np.ascontiguousarray(g["jobs%d" % k]).view(CU_MC_JOB_DTYPE)
        assert jobs.tobytes() == make_jobs(r, 24, w, h, cuw, cuh, nref).tobytes()  # the generator still draws the same cases
        cw, ch = cuw >> refs["ws"], cuh >> refs["hs"]
        for i in range(len(jobs)):
            e = [np.zeros(cuw * cuh, np.int16), np.zeros(cw * ch, np.int16), np.zeros(cw * ch, np.int16)]
            O.xo_mc_cu(ptr(tab), refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), cuw, cuh, bd, bd, idc, ptr(e[0]), ptr(e[1]), ptr(e[2]))
            for c in range(3 if idc else 1):
                assert np.array_equal(e[c], g["out%d_%d" % (k, c)][i]), (k, i, c)
