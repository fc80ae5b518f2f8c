"""Iterates tests/golden/me_v1.npz (results of the reference's own me_ipel_diamond) as case dicts."""
import os

import numpy as np

from _me_cases import PAD

ME_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_v1.npz")


def golden_cases():
    g = np.load(ME_GOLDEN)
    for t in (0, 1):
        org, ref = np.ascontiguousarray(g["org%d" % t]), np.ascontiguousarray(g["ref%d" % t])
        W, H = 128, 96
        for row, obi in zip(g["jobs%d" % t], g["org_bi%d" % t]):
            (S, bi, x, y, r0, r1, r2, r3, gx, gy, ix, iy, msr, sr, lam, fast, mot, bs_in, cost, mvx, mvy, beststep) = (int(v) for v in row)
            c = dict(org=org, ref=ref, s=W + 2 * PAD, x=x, y=y, S=S, bi=bi, min_clip=(-127, -127), max_clip=(W - 1, H - 1),
                     range=[r0, r1, r2, r3], gmvp=(gx, gy), mvi=(ix, iy), msr=msr, sr=sr, lambda_mv=lam, faststep=fast, mot_other=mot,
                     org_bi=np.ascontiguousarray(obi[:S * S]), beststep_in=bs_in)
            yield c, (cost, mvx, mvy, beststep)


def golden_spel_cases():
    g = np.load(ME_GOLDEN)
    for t in (0, 1):
        org, ref = np.ascontiguousarray(g["org%d" % t]), np.ascontiguousarray(g["ref%d" % t])
        W = 128
        for row, obi in zip(g["spel_jobs%d" % t], g["spel_org_bi%d" % t]):
            (S, bi, x, y, gx, gy, ix, iy, lam, mot, hc, qc, cost, mvx, mvy) = (int(v) for v in row)
            c = dict(org=org, ref=ref, s=W + 2 * PAD, x=x, y=y, S=S, bi=bi, gmvp=(gx, gy), mvi=(ix, iy), lambda_mv=lam, mot_other=mot,
                     org_bi=np.ascontiguousarray(obi[:S * S]), hpel_cnt=hc, qpel_cnt=qc)
            yield c, (cost, mvx, mvy)
