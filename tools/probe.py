#!/usr/bin/env python3
"""Developer probe: per-kernel-family timing of the hot-path pass on the GPU (not part of the product or the tests).
usage: python tools/probe.py [sad|all] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import xeve_amd
from xeve_amd import device as D
from xeve_amd.workload import N_LIST, N_PASS, HotPathPass

what = sys.argv[1] if len(sys.argv) > 1 else "sad"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
xeve_amd.init(0)
dev = torch.device("cuda:0")
wl = HotPathPass(3840, 2160, dev)


def timeit(fn, reps=reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = 0.0
for S in wl.sizes:
    lv = wl.lv[S]

    def me():
        for i, jobs in enumerate(lv["me_jobs"]):
            D.sad_jobs_dual(wl.org[0], wl.s_l, wl.ref[i % N_LIST][0], wl.ref_s1[i % N_LIST], wl.s_l, jobs, wl.cand_l, S, S, 10, out=lv["sad_out"])

    ms = timeit(me)
    by = lv["n"] * N_LIST * N_PASS * len(wl.pattern) * (4 * S * S + 4)
    tot += ms
    print("SAD %2dx%-2d  %7.3f ms/picture  %8.1f GB/s algorithmic" % (S, S, ms, by / ms / 1e6))
print("SAD total %.3f ms" % tot)
if what == "all":
    for ph, name in (("A", "A integer ME SAD"), ("B", "B half-pel MC+SAD"), ("C", "C merge MC+SSD"), ("D1", "D1 bi-pred MC+avg"),
                     ("D2", "D2 fused residual chain"), ("E", "E SATD gate")):
        print("phase %-26s %.3f ms" % (name, timeit(lambda: wl.run(only=ph))))
    ms = timeit(lambda: wl.run())
    print("full pass %.3f ms  -> %.1f pictures/s" % (ms, 1e3 / ms))

if what == "graph":
    for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160)):
        wl2 = HotPathPass(w, h, dev)
        eager = timeit(lambda: wl2.run(), 20)
        wl2.capture()
        rep = timeit(lambda: wl2.replay(), 20)
        print("%dx%d eager %.3f ms  graph %.3f ms" % (w, h, eager, rep))
if what == "me":
    import ctypes as C
    import numpy as np
    from xeve_amd import lib
    from xeve_amd.workload import PAD_L
    W, H = 3840, 2160
    s = W + 2 * PAD_L
    rng = np.random.default_rng(5)
    # moving smooth texture + noise: the search has something to find
    yy, xx = np.mgrid[0:H + 2 * PAD_L, 0:s].astype(np.float32)
    base = 512 + 300 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 150 * np.sin((xx + yy) / 41.0)
    org_np = np.clip(base + rng.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)
    ref_np = np.clip(np.roll(base, (9, -13), axis=(0, 1)) + rng.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)
    org, ref = torch.from_numpy(org_np).to(dev), torch.from_numpy(ref_np).to(dev)
    P = lib.MeParams(1 << 20, 1, 0, 0, 3, 64, 64, (C.c_int32 * 2)(-128, -128), (C.c_int32 * 2)(W - 1 + 128, H - 1 + 128), 0)
    for S in (8, 16, 32, 64):
        ys, xs = np.meshgrid(np.arange(H // S) * S, np.arange(W // S) * S, indexing="ij")
        n = xs.size
        jobs = np.zeros(n, dtype=lib.ME_JOB_DTYPE)
        jobs["x"], jobs["y"] = xs.ravel(), ys.ravel()
        mvp = rng.integers(-32, 33, size=(n, 2))
        cx, cy = np.clip(jobs["x"] + (mvp[:, 0] >> 2), -128, W + 127), np.clip(jobs["y"] + (mvp[:, 1] >> 2), -128, H + 127)
        jobs["range"] = np.stack([np.clip(cx - 64, -128, W + 127), np.clip(cy - 64, -128, H + 127), np.clip(cx + 64, -128, W + 127),
                                  np.clip(cy + 64, -128, H + 127)], axis=1)
        jobs["gmvp"] = np.stack([mvp[:, 0] + (jobs["x"] << 2), mvp[:, 1] + (jobs["y"] << 2)], axis=1)
        jobs["mvi"] = jobs["gmvp"]
        o0 = PAD_L * s + PAD_L
        f = lambda: D.me_ipel_diamond_jobs(org, o0, s, None, ref, o0, s, jobs, S.bit_length() - 1, 10, P)
        res = f()
        torch.cuda.synchronize()
        import time as _t
        t0 = _t.perf_counter()
        for _ in range(3):
            res = f()
        torch.cuda.synchronize()
        ms = (_t.perf_counter() - t0) / 3 * 1e3
        print("me_ipel_diamond %2dx%-2d  %7d jobs  %.3f ms (incl. job upload / result download)  mean |mv| %.1f qpel, beststep>2 in %.0f%%"
              % (S, S, n, ms, np.abs(res["mv"]).mean(), 100.0 * (res["beststep"] > 2).mean()))
if what == "epzs":
    import numpy as np, time as _t
    from xeve_amd import me
    from xeve_amd.workload import PAD_L
    W, H = 3840, 2160
    s = W + 2 * PAD_L
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H + 2 * PAD_L, 0:s].astype(np.float32)
    base = 512 + 300 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 150 * np.sin((xx + yy) / 41.0)
    org = torch.from_numpy(np.clip(base + rng.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)).to(dev)
    ref = torch.from_numpy(np.clip(np.roll(base, (9, -13), axis=(0, 1)) + rng.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)).to(dev)
    o0 = PAD_L * s + PAD_L
    tot = 0.0
    for S in (8, 16, 32, 64):
        ys, xs = np.meshgrid(np.arange(H // S) * S, np.arange(W // S) * S, indexing="ij")
        x, y = xs.ravel(), ys.ravel()
        mvp = rng.integers(-32, 33, size=(x.size, 2))
        f = lambda: me.epzs_search_device(org, o0, s, ref, o0, s, x, y, mvp, S.bit_length() - 1, 10, 1 << 20, 1, 64, 64, (-128, -128), (W + 127, H + 127), 4, 0)
        f(); torch.cuda.synchronize()
        t0 = _t.perf_counter(); cost, mv = f(); torch.cuda.synchronize(); ms = (_t.perf_counter() - t0) * 1e3
        tot += ms
        print("epzs %2dx%-2d %7d blocks %.2f ms  (mean |mv| %.1f qpel)" % (S, S, x.size, ms, np.abs(mv).mean()))
    print("epzs all levels, one list: %.2f ms" % tot)
