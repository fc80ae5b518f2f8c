#!/usr/bin/env python3
"""Does a kernel launched earlier in the process change the composed walk's step time?  Each setting in a fresh interpreter: first the named work (affine searches of one CU
size through xeve_hip_affine_me_jobs -- 128x128 takes 69 KB of dynamic LDS --, or nothing), then 8 GOPs of 1280x512, IDR + B, timed per picture.
usage: probe_after.py [none me16 me64 me128 ...]"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    what = sys.argv[2]
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    if what.startswith("me"):
        import _affine_me as M
        from test_affine_me import HipAffineMe

        n = int(what[2:])
        jobs, bi = M.make_jobs(n, n, 5)
        HipAffineMe().run(M.ref_pictures(), M.org_picture(), jobs, bi, n, n)
    torch.cuda.synchronize()
    W, H, F, G = 1280, 512, 2, 8
    fb = W * H * 3 // 2
    cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=8)
    enc = encode.BatchEncoder(cfg, G, F)
    d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device="cuda")
    for g in range(G):
        for f in range(F):
            enc.push(g, f, d[f * fb:(f + 1) * fb])
    enc.begin()
    per = enc.advance(0) // F
    out = {"first": what}
    for pic in range(F):
        t = time.perf_counter()
        enc.advance(per)
        enc.sync()
        out["ms_per_step_picture_%d" % pic] = round(1e3 * (time.perf_counter() - t) / per, 2)
    print(json.dumps(out), flush=True)
    sys.exit(0)

for what in sys.argv[1:] or ["none", "me16", "me64", "me128"]:
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", what], capture_output=True, text=True, env=os.environ)
    print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:], flush=True)
