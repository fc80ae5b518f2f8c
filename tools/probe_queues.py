#!/usr/bin/env python3
"""Does the composed walk's step time depend on how many OTHER streams the process holds?  (HIP maps streams onto a few hardware queues: the walk's main and side stream --
or the encoder's second-pass stream -- sharing one would serialise them.)  Each setting in a fresh interpreter: k idle streams created first, then 8 GOPs of 1280x512, IDR + B.
usage: probe_queues.py [k ...]"""
import json
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    k = int(sys.argv[2])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")  # (torch.cuda.Stream() hands out streams of a pool it creates at once: plain hipStreamCreateWithFlags here, one queue user each)
    buf = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    extra = []
    for _ in range(k):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(1024), st) == 0
        extra.append(st)
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
    total = enc.advance(0)
    per = total // F
    out = {"extra_streams": k}
    for pic in range(F):
        t = time.perf_counter()
        enc.advance(per)
        enc.sync()
        out["ms_per_step_picture_%d" % pic] = round(1e3 * (time.perf_counter() - t) / per, 2)
    print(json.dumps(out), flush=True)
    sys.exit(0)

for k in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7]:
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(k)], capture_output=True, text=True, env=os.environ)
    print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
