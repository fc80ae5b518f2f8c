#!/usr/bin/env python3
"""Writes tests/golden/me_v1.npz: inputs and the results of the REFERENCE's static me_ipel_diamond
(src_base/xeve_pinter.c:363-551, via oracle/ref_me_driver.c) for seeded jobs on two plane pairs.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _me_cases import make_job, make_planes, make_spel_job  # noqa: E402
from test_me_oracle_vs_ref import run_ref, run_ref_spel  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "me_v1.npz")
r = np.random.default_rng(424242)
d = {}
for t, textured in enumerate((False, True)):
    pl = make_planes(r, textured, W=128, H=96)
    d["org%d" % t], d["ref%d" % t] = pl["org"], pl["ref"]
    rows = []
    bis = []
    for k in range(48):
        S, bi = int(r.choice([8, 16, 32, 64])), int(r.choice([0, 0, 0, 1, 2]))
        c = make_job(r, pl, S, bi)
        cost, mvx, mvy, beststep, _ = run_ref(c)
        rows.append([S, bi, c["x"], c["y"], *c["range"], *c["gmvp"], *c["mvi"], c["msr"], c["sr"], c["lambda_mv"], c["faststep"], c["mot_other"],
                     c["beststep_in"], cost, mvx, mvy, beststep])
        bis.append(np.pad(c["org_bi"], (0, 4096 - len(c["org_bi"]))))
    d["jobs%d" % t] = np.array(rows, np.int64)
    d["org_bi%d" % t] = np.array(bis, np.int16)
    rows, bis = [], []
    for k in range(32):  # me_spel_pattern
        S, bi = int(r.choice([8, 16, 32, 64])), int(r.choice([0, 0, 1]))
        c = make_spel_job(r, pl, S, bi)
        cost, mvx, mvy, _ = run_ref_spel(c)
        rows.append([S, bi, c["x"], c["y"], *c["gmvp"], *c["mvi"], c["lambda_mv"], c["mot_other"], c["hpel_cnt"], c["qpel_cnt"], cost, mvx, mvy])
        bis.append(np.pad(c["org_bi"], (0, 4096 - len(c["org_bi"]))))
    d["spel_jobs%d" % t] = np.array(rows, np.int64)
    d["spel_org_bi%d" % t] = np.array(bis, np.int16)
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT))
