#!/usr/bin/env python3
"""Writes tests/golden/cand_v1.npz: CU positions on seeded per-unit maps and the candidates the reference derives for them (xeve_get_avail_inter +
xeve_get_motion + the collocated vector of xeve_get_mv_dir, via oracle/ref_rdo_driver.c).  The maps are regenerated from the seed.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _cand_golden import CASES, W_SCU, H_SCU, positions  # noqa: E402
from _inter_cases import make_maps  # noqa: E402
from _libs import INTER_JOB_DTYPE, ptr, ref_cand  # noqa: E402

R = ref_cand()
d = {}
for k, (seed, slice_type, tiles, lw) in enumerate(CASES):
    r = np.random.default_rng(seed)
    map_scu, tidx, map_mv, c0, c1 = make_maps(r, W_SCU, H_SCU, tiles)
    jobs = positions(r, lw)
    for i in range(len(jobs)):
        R.refdrv_inter_candidates(ptr(map_scu), ptr(tidx), ptr(map_mv), ptr(c0), ptr(c1), W_SCU, H_SCU, lw, lw, slice_type, ptr(jobs[i:i + 1]))
    d["jobs%d" % k] = jobs.view(np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cand_v1.npz")
np.savez_compressed(out, **d)
print("wrote", out, os.path.getsize(out))
