#!/usr/bin/env python3
"""Developer probe: (SPLIT, UNROLL) variants of k_sad_sq per block size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xeve_amd
from xeve_amd import device as D
from xeve_amd.workload import N_LIST, HotPathPass
xeve_amd.init(0)
wl = HotPathPass(3840, 2160, torch.device("cuda:0"))
names = {8: ["1x6", "1x12", "2x6", "1x6"], 16: ["2x8", "2x12", "4x6", "1x16"], 32: ["2x8", "2x12", "4x8", "4x6"], 64: ["4x4", "4x6", "8x4", "6x4"]}
for S in wl.sizes:
    lv = wl.lv[S]
    for tune in range(4):
        def me():
            for i, jobs in enumerate(lv["me_jobs"]):
                D.sad_jobs_dual(wl.org[0], wl.s_l, wl.ref[i % N_LIST][0], wl.ref_s1[i % N_LIST], wl.s_l, jobs, wl.cand_l, S, S, 10, out=lv["sad_out"], tune=tune)
        me(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): me()
        e1.record(); torch.cuda.synchronize()
        print("S=%2d split x unroll %-4s %.3f ms" % (S, names[S][tune], e0.elapsed_time(e1) / 10))
