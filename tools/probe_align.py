#!/usr/bin/env python3
"""Developer probe: sensitivity of the SAD kernel to the x-alignment of the reference rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xeve_amd
from xeve_amd import device as D
from xeve_amd.workload import PAD_L, diamond_pattern
xeve_amd.init(0)
dev = torch.device("cuda:0")
W, H = 3840, 2160
s = W + 2 * PAD_L
g = torch.Generator(device=dev).manual_seed(1)
org = torch.randint(0, 1024, (H + 2 * PAD_L, s), generator=g, device=dev, dtype=torch.int16)
ref = torch.randint(0, 1024, (H + 2 * PAD_L, s), generator=g, device=dev, dtype=torch.int16)
pat = diamond_pattern()
rng = np.random.default_rng(0)
ref_s1 = D.plane_shift1(ref)
def run(S, xalign, label, mode=3):
    nx, ny = W // S, H // S
    ys, xs = np.meshgrid(np.arange(ny) * S, np.arange(nx) * S, indexing="ij")
    ys, xs = ys.ravel(), xs.ravel()
    n = len(xs)
    off = (PAD_L + ys) * s + PAD_L + xs
    mvx = (rng.integers(-48, 49, n) // xalign) * xalign
    mvy = rng.integers(-48, 49, n)
    jobs = D.make_jobs(off, off + mvy * s + mvx, dev)
    cand = torch.tensor([dy * s + (dx // xalign) * xalign for dx, dy in pat], dtype=torch.int32, device=dev)
    out = torch.empty((n, len(pat)), dtype=torch.int32, device=dev)
    if mode == 1:
        f = lambda: D.sad_jobs_dual(org, s, ref, ref_s1, s, jobs, cand, S, S, 10, out=out)
    else:
        f = lambda: D.sad_jobs(org, s, ref, s, jobs, cand, S, S, 10, out=out, mode=mode)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("S=%2d %-12s %.3f ms  %.1f GB/s alg" % (S, label, ms, n * len(pat) * (4 * S * S + 4) / ms / 1e6))
for S in (8, 16, 32, 64):
    run(S, 1, "plain any", 3)
    run(S, 1, "dual any", 1)
    run(S, 1, "funnel any", 2)
    run(S, 2, "plain x%2", 3)
    run(S, 2, "funnel x%2", 2)
