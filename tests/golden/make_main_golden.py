#!/usr/bin/env python3
"""Writes tests/golden/main_v1.npz: the outputs of the Main-profile reference library's OWN tables (xevem_tbl_dmvr_mc_l / _c, xevem_tbl_bl_mc_l,
xeve_tbl_tx, xeve_tbl_itx; plain-C variants of oracle/_ref/libxevem_ref.so) on the case list of tests/_main_cases.py, plus a checksum of the
seeded inputs (so a change of the generator shows up as such, not as a parity failure).  Build container only."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _main_cases import GOLDEN, REF_NAMES, TableMain, input_checksum, ref_main_lib, run_all  # noqa: E402

out = np.concatenate(run_all(TableMain(ref_main_lib(), REF_NAMES["c"])))
np.savez_compressed(GOLDEN, out=out, inputs_crc=np.array(input_checksum(), np.uint32), out_crc=np.array(zlib.crc32(out.tobytes()), np.uint32))
print(GOLDEN, out.size, "samples", os.path.getsize(GOLDEN), "bytes")
