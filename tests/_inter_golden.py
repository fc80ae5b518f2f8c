"""Iterator over tests/golden/inter_v1.npz (reference results of xeve_pinter_analyze_cu); pictures / states regenerated from the seed."""
import os

import numpy as np

from _inter_cases import make_inter_jobs, make_inter_params, make_inter_picture
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, sbac_from_golden
from _rdo_cases import states

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inter_v1.npz")
N_JOBS = 20
# seed, w, h, bit depth, reference pictures per list, chroma_format_idc, slice type (0 B, 1 P), log2 CU size, skip_th
CASES = [(901, 128, 96, 10, 2, 1, 0, 3, 0.0), (902, 128, 96, 10, 2, 1, 0, 4, 0.0), (903, 128, 96, 10, 2, 1, 0, 5, 0.0), (904, 128, 128, 10, 2, 1, 0, 6, 0.0),
         (905, 128, 64, 10, 2, 1, 1, 4, 0.0), (906, 96, 64, 8, 1, 1, 0, 4, 0.0), (907, 64, 64, 10, 2, 0, 0, 3, 0.0), (908, 192, 128, 10, 3, 1, 0, 4, 0.0),
         (909, 128, 96, 10, 2, 1, 0, 4, 6.0), (910, 128, 128, 10, 4, 1, 1, 5, 0.0)]


def golden():
    g = np.load(GOLD)
    for k, (seed, w, h, bd, nref, idc, st_type, lw, skip_th) in enumerate(CASES):
        r = np.random.default_rng(seed)
        refs, org = make_inter_picture(r, w, h, bd, nref, idc, st_type)
        st = states(r, 6)
        P = make_inter_params(r, lw, w, h, bd, nref, idc, st_type, refs, skip_th)
        jobs = make_inter_jobs(r, N_JOBS, w, h, 1 << lw, len(st), refs, st_type)
        assert bytes(P) == np.ascontiguousarray(g["params%d" % k]).tobytes() and jobs.tobytes() == np.ascontiguousarray(g["jobs%d" % k]).tobytes()
        yield dict(refs=refs, org=org, states=st, P=P, jobs=jobs, res=np.ascontiguousarray(g["res%d" % k]).view(INTER_RESULT_DTYPE),
                   best=sbac_from_golden(g["best%d" % k], st[jobs["sbac"]]), coef=[g["coef%d_%d" % (k, c)] for c in range(3)],
                   rec=[g["rec%d_%d" % (k, c)] for c in range(3)], idc=idc, lw=lw, slice_type=st_type)
