"""Iterator over tests/golden/skip_v1.npz (reference results of xeve_analyze_skip); pictures / states regenerated from the seed."""
import os

import numpy as np

from _libs import SBAC_DTYPE, sbac_from_golden, SKIP_RESULT_DTYPE
from _rdo_cases import make_params, make_picture, make_skip_jobs, states

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skip_v1.npz")
N_CASES = 7


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        seed, w, h, bd, nref, idc, st_type, lw, lh, ncand = (int(v) for v in g["p%d" % k])
        r = np.random.default_rng(seed)
        refs, org = make_picture(r, w, h, bd, nref, idc)
        st = states(r, 6)
        p = make_params(r, lw, lh, w, h, bd, nref, idc, st_type)
        jobs = make_skip_jobs(r, 24, w, h, 1 << lw, 1 << lh, len(st), ncand)
        assert bytes(p) == np.ascontiguousarray(g["params%d" % k]).tobytes() and jobs.tobytes() == np.ascontiguousarray(g["jobs%d" % k]).tobytes()
        yield dict(refs=refs, org=org, states=st, p=p, jobs=jobs, res=np.ascontiguousarray(g["res%d" % k]).view(SKIP_RESULT_DTYPE),
                   best=sbac_from_golden(g["best%d" % k], st[jobs["sbac"]]), pred=[g["pred%d_%d" % (k, c)] for c in range(3)], idc=idc, lw=lw, lh=lh,
                   slice_type=st_type, ncand=ncand)
