#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Writes tests/golden/alf_v1.npz: what the UNMODIFIED reference's own ALF sample kernels (oracle/_ref/libxevem_ref.so: alf_derive_classification_blk,
alf_filter_blk_7 / _5, xeve_alf_get_blk_stats, alf_copy_and_extend) produce for the seeded cases of tests/_alf.py -- classifier planes, filtered planes, the per-class
correlation records.  Inputs are regenerated from the seeds by the tests; only outputs are stored."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _alf  # noqa: E402

R = _alf.RefAlf()
out = {}
for name in _alf.CASES:
    for k, v in _alf.run_case(R, name).items():
        out[name + "/" + k] = v
np.savez_compressed(_alf.GOLDEN, **out)
print(len(out), "arrays,", os.path.getsize(_alf.GOLDEN), "bytes")
