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
