"""tests/golden/cand_v1.npz: reference results of the candidate derivation; maps and positions regenerated from the seed."""
import os

import numpy as np

from _inter_cases import make_maps
from _libs import INTER_JOB_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cand_v1.npz")
W_SCU, H_SCU = 48, 32
# seed, slice type, tiles, log2 CU size
CASES = [(1001, 0, 1, 3), (1002, 0, 1, 4), (1003, 0, 1, 5), (1004, 0, 1, 6), (1005, 1, 1, 3), (1006, 1, 1, 5), (1007, 0, 2, 3), (1008, 0, 2, 4)]


def positions(r, lw, n=120):
    s = 1 << (lw - 2)
    j = np.zeros(n, INTER_JOB_DTYPE)
    j["x"] = r.integers(0, W_SCU // s, size=n) * s * 4
    j["y"] = r.integers(0, H_SCU // s, size=n) * s * 4
    return j


def golden():
    g = np.load(GOLD)
    for k, (seed, slice_type, tiles, lw) in enumerate(CASES):
        r = np.random.default_rng(seed)
        maps = make_maps(r, W_SCU, H_SCU, tiles)
        jobs = positions(r, lw)
        exp = np.ascontiguousarray(g["jobs%d" % k]).view(INTER_JOB_DTYPE)
        assert np.array_equal(jobs["x"], exp["x"]) and np.array_equal(jobs["y"], exp["y"])
        yield dict(maps=maps, jobs=jobs, exp=exp, slice_type=slice_type, tiles=tiles, lw=lw)
