"""Iterator over tests/golden/rdo_v1.npz (reference results of pinter_residue_rdo); pictures / states regenerated from the seed."""
import os

import numpy as np

from _libs import RDO_JOB_DTYPE, SBAC_DTYPE, sbac_from_golden, RdoParams
from _rdo_cases import make_jobs, make_params, make_picture, states

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdo_v1.npz")


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        seed, w, h, bd, nref, idc, st_type, lw, lh = (int(v) for v in g["p%d" % k])
        r = np.random.default_rng(seed)
        refs, org = make_picture(r, w, h, bd, nref, idc)
        st = states(r, 6)
        p = make_params(r, lw, lh, w, h, bd, nref, idc, st_type)
        jobs = make_jobs(r, 24, w, h, 1 << lw, 1 << lh, nref, len(st), st_type)
        assert bytes(p) == np.ascontiguousarray(g["params%d" % k]).tobytes() and jobs.tobytes() == np.ascontiguousarray(g["jobs%d" % k]).tobytes()
        yield dict(refs=refs, org=org, states=st, p=p, jobs=jobs, cost=g["cost%d" % k], nnz=g["nnz%d" % k],
                   best=sbac_from_golden(g["best%d" % k], st[jobs["sbac"]]), coef=[g["coef%d_%d" % (k, c)] for c in range(3)], idc=idc, lw=lw, lh=lh)
