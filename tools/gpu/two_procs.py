#!/usr/bin/env python3
"""Experiment: do two batches driven by two PROCESSES (own HIP runtime each) go faster than by two threads of one process?  usage: two_procs.py W H GOPS FRAMES NPROC"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(i, W, H, G, F, bar, out):
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    dev = torch.device("cuda", 0)
    fb = W * H * 3 // 2
    cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=8)
    enc = encode.BatchEncoder(cfg, G, F)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + i)
    for g in range(G):
        d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device=dev, generator=gen)
        for f in range(F):
            enc.push(g, f, d[f * fb:(f + 1) * fb])
    enc.begin()
    total = enc.advance(0)
    enc.sync()
    bar.wait()
    t0 = time.time()
    left = total
    while left > 0:
        left = enc.advance(1 << 20)
    enc.sync()
    t1 = time.time()
    out.put((i, t0, t1, total, enc.stats()["step_seconds"]))
    enc.close()


if __name__ == "__main__":
    W, H, G, F, N = (int(x) for x in sys.argv[1:6])
    ctx = mp.get_context("spawn")
    bar, out = ctx.Barrier(N), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(i, W, H, G, F, bar, out)) for i in range(N)]
    for p in ps:
        p.start()
    res = [out.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
    t0, t1 = min(r[1] for r in res), max(r[2] for r in res)
    print("procs", N, "gops each", G, "frames/s", round(N * G * F / (t1 - t0), 2), "wall", round(t1 - t0, 2), "per proc", [(round(r[2] - r[1], 2), r[3], round(r[4], 2)) for r in res], flush=True)
