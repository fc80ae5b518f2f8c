#!/usr/bin/env python3
"""Writes tests/golden/enc_v1.json from the UNMODIFIED reference application (oracle/_ref/xeveb_app).  Build container only.
  "plans":   per (options, frames) the table the application prints per coded picture -- POC, temporal id, slice type, QP, first reference picture of each list;
  "batches": per case of tests/_enc.py BATCH_CASES(_REAL) the md5 + size of every GOP's bitstream, each GOP coded as a run of its own (--seek g * F --frames F).
usage: make_enc_golden.py [plans] [case names ...] -- without arguments everything is (re)made."""
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _e2e import make_yuv  # noqa: E402
from _enc import BATCH_CASES, BATCH_CASES_REAL, DEPTH10_CASES, GOLDEN, HEADER_OPTION_CASES, HOST_PINNED_CASES, PLACEBO_BATCH_CASES, PLAN_GRID, PRESET_REAL_CASES, SLOW_BATCH_CASES, app_args_and_env, md5, widen10  # noqa: E402
from _libs import REF_APP  # noqa: E402

ROW = re.compile(r"^(?:\[.*?\] )*(\d+)\s+(\d+)\s+\((.)\)\s+(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+\d+\s+\d+\s*(.*)$")


def app(yuv, out, w, h, frames, cli, threads, seek=0):
    args, pin = app_args_and_env(cli)  # (options the application cannot parse travel in the environment: oracle/ref_param_pin.c)
    cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-m", str(threads), "-v", "3", "-o", out] + args
    if seek:
        cmd += ["--seek", str(seek)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **pin))
    assert p.returncode == 0, (cmd, p.stdout[-1500:], p.stderr[-2000:])
    return p.stdout


only = sys.argv[1:]
out = json.load(open(GOLDEN)) if only and os.path.exists(GOLDEN) else {"plans": [], "batches": {}}
with tempfile.TemporaryDirectory() as d:
    if not only or "plans" in only:
        out["plans"] = []
        yuv = os.path.join(d, "p.yuv")
        make_yuv(yuv, 64, 64, 40, 1)
        for cli, counts in PLAN_GRID:
            for n in counts:
                rows = []
                for line in app(yuv, os.path.join(d, "p.evc"), 64, 64, n, cli, 1).replace("[ ", "\n[ ").split("\n"):
                    m = re.search(r"(\d+)\s+(\d+)\s+\((.)\)\s+(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+\d+\s+\d+\s*(.*)$", line)
                    if m and "frame/sec" not in m.group(0)[:m.start(3) - m.start(0)]:
                        l0, l1 = re.search(r"\[L0 (\d+)", m.group(5)), re.search(r"\[L1 (\d+)", m.group(5))
                        rows.append([int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4)), int(l0.group(1)) if l0 else -1, int(l1.group(1)) if l1 else -1])
                assert len(rows) == n, (cli, n, rows)
                out["plans"].append({"cli": cli, "frames": n, "rows": rows})
        print("plans:", len(out["plans"]))
    for name, (w, h, gops, frames, seed, cli, threads) in list(BATCH_CASES.items()) + list(BATCH_CASES_REAL.items()) + list(DEPTH10_CASES.items()) + list(HOST_PINNED_CASES.items()) + list(HEADER_OPTION_CASES.items()) + list(SLOW_BATCH_CASES.items()) + list(PLACEBO_BATCH_CASES.items()) + list(PRESET_REAL_CASES.items()):
        if only and name not in only:
            continue
        yuv, evc = os.path.join(d, name + ".yuv"), os.path.join(d, name + ".evc")
        make_yuv(yuv, w, h, gops * frames, seed)
        if name in DEPTH10_CASES:  # (the application reads 16-bit samples with -d 10)
            wide = widen10(open(yuv, "rb").read())
            open(yuv, "wb").write(wide)
        per = []
        for g in range(gops):
            app(yuv, evc, w, h, frames, cli, threads, seek=g * frames)
            b = open(evc, "rb").read()
            per.append({"md5": md5(b), "bytes": len(b)})
        out["batches"][name] = {"w": w, "h": h, "gops": gops, "frames": frames, "seed": seed, "cli": cli, "threads": threads, "per_gop": per}
        if gops > 1 and (name in BATCH_CASES or name in DEPTH10_CASES or name in HOST_PINNED_CASES or name in HEADER_OPTION_CASES or name in SLOW_BATCH_CASES or name in PLACEBO_BATCH_CASES):  # the whole sequence in ONE run of the reference: what the concatenated per-GOP bitstreams must be
            app(yuv, evc, w, h, gops * frames, cli, threads)
            b = open(evc, "rb").read()
            out["batches"][name]["whole"] = {"md5": md5(b), "bytes": len(b)}
        print(name, per, out["batches"][name].get("whole"))
json.dump(out, open(GOLDEN, "w"), indent=1)
