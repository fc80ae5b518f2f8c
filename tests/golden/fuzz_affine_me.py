#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs oracle/_ref/libref_affine_me.so).  Random affine gradient searches -- wilder start vectors and predictors than the goldens', random lambdas
and list lengths -- through the oracle's restatement, the kernel's scalar steps compiled for the host (affine_core.h) and the reference's own pinter_affine_me_gradient: all
three must agree on vectors and value.  usage: fuzz_affine_me.py [seeds, default 30]   (30 seeds = 5760 searches, ~40 s)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _affine_me as M  # noqa: E402

pics, org = M.ref_pictures(), M.org_picture()
O, R, H = M.OracleAffineMe(), M.RefAffineMe(1), M.HostAffineMe()
bad = n = 0
t0 = time.time()
for seed in range(3000, 3000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 30)):
    for (w, h) in M.SIZES:
        jobs, ob = M.make_jobs(w, h, seed * 13 + w + h, n=24)
        g = np.random.default_rng(seed)
        for j in jobs:
            if g.integers(0, 3) == 0:
                j["mv"] += g.integers(-200, 201, size=(3, 2)).astype(np.int16)
            if g.integers(0, 5) == 0:
                j["mvp"] += g.integers(-4000, 4001, size=(3, 2)).astype(np.int16)
        lam, nr = int(g.integers(1000, 6000000)), int(g.integers(3, 6))
        a, b, c = O.run(pics, org, jobs, ob, w, h, lam, nr), R.run(pics, org, jobs, ob, w, h, lam, nr), H.run(pics, org, jobs, ob, w, h, lam, nr)
        n += len(jobs)
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad += 1
            print("oracle differs from the reference", seed, w, h)
        k = [3 if v == 3 else 2 for v in jobs["vertex_num"]]
        if not all(np.array_equal(c[0][i][:k[i]], b[0][i][:k[i]]) and c[1][i] == b[1][i] for i in range(len(jobs))):
            bad += 1
            print("affine_core.h differs from the reference", seed, w, h)
print("searches", n, "bad", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
