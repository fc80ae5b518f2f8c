#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Writes tests/golden/affine_v1.npz from the UNMODIFIED reference's own xeve_affine_mc (oracle/_ref/libref_affine.so = oracle/ref_affine_driver.c
around oracle/_ref/libxevem_ref.so): per CU size of tests/_affine.py SIZES the md5 of every job's prediction planes and the path it took (sub-block width / height, the
memory-bandwidth condition).  Inputs are regenerated from the seeds by the tests."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _affine as A  # noqa: E402

R = A.RefAffine()
pics = A.ref_pictures(1)
out = {}
for (w, h) in A.SIZES:
    Y, U, V, path = R.run(pics, A.make_jobs(w, h, 7 + w + h), w, h)
    out["%dx%d/md5" % (w, h)], out["%dx%d/path" % (w, h)] = A.digests(Y, U, V), path
np.savez_compressed(A.GOLDEN, **out)
print(len(out), "arrays,", os.path.getsize(A.GOLDEN), "bytes")
