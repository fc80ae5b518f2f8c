#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Writes tests/golden/affine_me_v1.npz from the UNMODIFIED reference's own pinter_affine_me_gradient (oracle/_ref/libref_affine_me.so =
oracle/ref_affine_me_driver.c, which compiles src_main/xevem_pinter.c in place, around oracle/_ref/libxevem_ref.so; the SSE kernels the application's build runs): per CU size
of tests/_affine_me.py SIZES the control points found and the value returned for every job.  Inputs are regenerated from the seeds by the tests."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _affine_me as M  # noqa: E402

R = M.RefAffineMe(1)
pics, org = M.ref_pictures(), M.org_picture()
out = {}
for (w, h) in M.SIZES:
    jobs, org_bi = M.make_jobs(w, h, 11 + w + h)
    out["%dx%d/mv" % (w, h)], out["%dx%d/cost" % (w, h)] = R.run(pics, org, jobs, org_bi, w, h)
np.savez_compressed(M.GOLDEN, **out)
print(len(out), "arrays,", os.path.getsize(M.GOLDEN), "bytes")
