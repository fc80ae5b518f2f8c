"""tests/golden/me_epzs_v1.npz: reference results of pinter_me_epzs (every branch: plain, me_raster, me_ipel_refinement, bi); planes and jobs are regenerated
from the seed."""
import os

import numpy as np

from _me_cases import make_epzs_job, make_planes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_epzs_v1.npz")
N = 120


def cases():
    r = np.random.default_rng(515151)
    out = []
    for it in range(N):
        pl = make_planes(r, textured=it % 5 != 0)
        S, bi = int(r.choice([8, 16, 32, 64])), int(r.choice([0, 0, 0, 1]))
        c = make_epzs_job(r, pl, S, bi)
        c["raster"], c["refi"] = int(r.random() < 0.6), int(r.integers(0, 2))
        if r.random() < 0.35:
            c["hpel_cnt"], c["qpel_cnt"] = 0, 0
        if r.random() < 0.7:
            c["mvp"] = (int(r.integers(-160, 161)), int(r.integers(-160, 161)))
        out.append(c)
    return out
