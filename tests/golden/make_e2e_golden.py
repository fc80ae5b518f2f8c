#!/usr/bin/env python3
"""Writes tests/golden/e2e_v1.json: md5 + size of the bitstreams the UNMODIFIED reference app (oracle/_ref/xeveb_app,
plain CPU dispatch) produces for the seeded synthetic clips of tests/_e2e.py.  Build container only."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _e2e import CASES, PLACEBO_CASES, REAL_CASES, SLOW_CASES, make_yuv, run_app  # noqa: E402

# usage: make_e2e_golden.py [case names ...] -- without names every case is (re)made; with names only those, merged into the existing file
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_v1.json")
only = sys.argv[1:]
out = json.load(open(PATH)) if only else {}
with tempfile.TemporaryDirectory() as d:
    for name, (w, h, n, seed, extra) in list(CASES.items()) + list(REAL_CASES.items()) + list(SLOW_CASES.items()) + list(PLACEBO_CASES.items()):
        if only and name not in only:
            continue
        yuv = os.path.join(d, name + ".yuv")
        make_yuv(yuv, w, h, n, seed)
        md5, size, _ = run_app(yuv, os.path.join(d, name + ".evc"), w, h, n, extra)
        out[name] = {"md5": md5, "bytes": size, "w": w, "h": h, "frames": n, "seed": seed, "cli": extra}
        print(name, md5, size)
json.dump(out, open(PATH, "w"), indent=1)
