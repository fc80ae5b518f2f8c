#!/usr/bin/env python3
"""Writes tests/golden/e2e_main_v1.json: md5 + size of the bitstreams the UNMODIFIED reference app on the Main-profile library
(oracle/_ref/xevem_app, plain CPU dispatch) produces for the seeded clips of tests/_e2e.py MAIN_CASES.  Build container only."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _e2e import MAIN_ALF_CASES, MAIN_CASES, make_yuv, run_app_main  # noqa: E402

out = {}
with tempfile.TemporaryDirectory() as d:
    for name, (w, h, n, seed, extra) in list(MAIN_CASES.items()) + list(MAIN_ALF_CASES.items()):
        yuv = os.path.join(d, name + ".yuv")
        make_yuv(yuv, w, h, n, seed)
        md5, size, _ = run_app_main(yuv, os.path.join(d, name + ".evc"), w, h, n, extra)
        assert (md5, size) == run_app_main(yuv, os.path.join(d, name + "2.evc"), w, h, n, extra)[:2]  # deterministic
        out[name] = {"md5": md5, "bytes": size, "w": w, "h": h, "frames": n, "seed": seed, "cli": extra}
        print(name, md5, size)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_main_v1.json"), "w"), indent=1)
