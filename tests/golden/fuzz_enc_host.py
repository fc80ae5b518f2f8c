#!/usr/bin/env python3
"""Build container only: random option sets of the supported space, each coded by the UNMODIFIED reference application (oracle/_ref/xeveb_app) and by the product's frame
loop on the CPU harness (tests/_enc.py encode_cpu); any difference is printed.  usage: fuzz_enc_host.py [count] [seed] [presets, comma-separated: default fast,medium,slow,placebo]"""
import os
import random
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _enc  # noqa: E402
from _e2e import make_yuv  # noqa: E402
from _libs import REF_APP  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PRESETS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fast", "medium", "slow", "placebo"]
bad = 0
with tempfile.TemporaryDirectory() as d:
    for it in range(count):
        w, h = rnd.choice([(64, 64), (128, 64), (72, 40), (136, 72), (64, 136), (128, 136), (200, 72), (136, 264)])
        frames = rnd.choice([1, 2, 3, 5, 8, 9, 12, 17])
        bf = rnd.choice([0, 1, 3, 7, 15])
        closed = rnd.random() < 0.5
        preset = rnd.choice(PRESETS)
        cli = ["--preset", preset, "-b", str(bf), "-q", str(rnd.choice([18, 27, 32, 37, 45]))]
        if closed:
            cli += ["--closed-gop", "-I", str(rnd.choice([1, 2, 4, 5, 8, 12, 16]))]
        else:
            k = rnd.choice([0, 1, 2, 4]) * (bf + 1)
            cli += ["-I", str(k)]
        if rnd.random() < 0.3:
            cli += ["--ref", str(rnd.choice([1, 2, 3]))]
        if rnd.random() < 0.35:  # (options the application cannot parse: they reach the library through oracle/ref_param_pin.c)
            cli += ["--inter-slice-type", "1"]
        if rnd.random() < 0.3 and preset in ("fast", "medium"):  # (slow / placebo: the loop filter's share of the chroma distortions is coded for offsets of 0)
            cli += ["--qp-cb-offset", str(rnd.randint(-12, 12)), "--qp-cr-offset", str(rnd.randint(-12, 12))]
        if rnd.random() < 0.2:
            cli += ["--info", "0"]
        if rnd.random() < 0.2:
            cli += ["--level-idc", str(rnd.choice([10, 30, 41, 51, 62]))]
        depth10 = rnd.random() < 0.25
        if depth10:
            cli += ["-d", "10"]
        threads = rnd.choice([1, 1, 2, 3, 5, 8]) if w > 64 else 1  # (one CTU wide with several threads: the reference itself is not deterministic, enc_plan.h)
        seed = rnd.choice([11, 5011, 6011])
        yuv, evc = os.path.join(d, "a.yuv"), os.path.join(d, "a.evc")
        make_yuv(yuv, w, h, frames, seed)
        data = open(yuv, "rb").read()
        if depth10:
            data = _enc.widen10(data)
            open(yuv, "wb").write(data)
        args, pin = _enc.app_args_and_env(cli)
        p = subprocess.run([REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-m", str(threads), "-v", "0", "-o", evc] + args, capture_output=True,
                           text=True, env=dict(os.environ, **pin))
        if p.returncode != 0:
            print("ref refuses", w, h, frames, threads, cli)
            continue
        ref = open(evc, "rb").read()
        try:
            cfg = _enc.config(w, h, cli, threads)
            got = _enc.encode_cpu(cfg, [data], frames)[0]
        except Exception as e:  # a configuration the library refuses
            print("refused", w, h, frames, threads, cli, str(e)[:100])
            continue
        if got != ref:
            bad += 1
            print("DIFF", w, h, frames, threads, seed, cli, len(got), len(ref))
print("done:", count, "cases,", bad, "different")
