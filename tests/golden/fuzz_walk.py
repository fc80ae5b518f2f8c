#!/usr/bin/env python3
"""Random option sets of the supported space, each coded twice by the product's frame loop on the CPU harness (tests/_enc.py encode_cpu): every CTU decided by the pinned
oracle, and every CTU decided by the HOST side of the fused walk (xeve_amd/csrc/walk.h, the code libxeve_hip.so runs as one kernel per step).  Any difference between the
two bitstreams is printed.  No reference needed (runs anywhere the CPU suite runs).  usage: fuzz_walk.py [count] [seed] [big|small] [presets, comma-separated: default fast,medium,slow,placebo]"""
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _enc  # noqa: E402
from _e2e import make_yuv  # noqa: E402

BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PRESETS = sys.argv[4].split(",") if len(sys.argv) > 4 else ["fast", "medium", "slow", "placebo"]
h_ = _enc.harness()
bad = done = 0
with tempfile.TemporaryDirectory() as d:
    for it in range(count):
        w, h = rnd.choice([(64, 64), (128, 64), (72, 40), (136, 72), (64, 136), (128, 136), (200, 72), (136, 264)] + ([(264, 200), (320, 192), (256, 256), (352, 288)] if BIG else []))
        frames = rnd.choice([1, 2, 3, 5, 8, 9, 12, 17])
        bf = rnd.choice([0, 1, 3, 7, 15])
        preset = rnd.choice(PRESETS)
        cli = ["--preset", preset, "-b", str(bf), "-q", str(rnd.choice([18, 27, 32, 37, 45]))]
        if rnd.random() < 0.5:
            cli += ["--closed-gop", "-I", str(rnd.choice([1, 2, 4, 5, 8, 12, 16]))]
        else:
            cli += ["-I", str(rnd.choice([0, 1, 2, 4]) * (bf + 1))]
        if rnd.random() < 0.3:
            cli += ["--ref", str(rnd.choice([1, 2, 3]))]
        if rnd.random() < 0.3:
            cli += ["--inter-slice-type", "1"]
        if rnd.random() < 0.3 and preset in ("fast", "medium"):  # (slow / placebo: the loop filter's share of the chroma distortions is coded for offsets of 0)
            cli += ["--qp-cb-offset", str(rnd.randint(-12, 12)), "--qp-cr-offset", str(rnd.randint(-12, 12))]
        depth10 = rnd.random() < 0.2
        if depth10:
            cli += ["-d", "10"]
        threads = rnd.choice([1, 1, 2, 3, 5, 8]) if w > 64 else 1
        seed = rnd.choice([11, 5011, 6011])
        yuv = os.path.join(d, "a.yuv")
        make_yuv(yuv, w, h, frames, seed)
        data = open(yuv, "rb").read()
        if depth10:
            data = _enc.widen10(data)
        try:
            cfg = _enc.config(w, h, cli, threads)
            h_.xo_encode_use_walk(0)
            want = _enc.encode_cpu(cfg, [data], frames)[0]
        except Exception as e:  # a configuration the library refuses
            print("refused", w, h, frames, threads, cli, str(e)[:100], flush=True)
            continue
        h_.xo_encode_use_walk(1)
        try:
            got = _enc.encode_cpu(cfg, [data], frames)[0]
        finally:
            h_.xo_encode_use_walk(0)
        done += 1
        if got != want:
            bad += 1
            print("DIFF", w, h, frames, threads, seed, cli, len(got), len(want), flush=True)
print("done:", done, "cases coded,", bad, "different", flush=True)
