#!/usr/bin/env python3
"""Is the encoder slowed by its OWN second stream's copies?  (A live extra stream that has moved pinned memory makes a process issue later launches slower: profiles/r07n.)
All-intra run (every picture the same work): ms per lockstep step over the FIRST HALF of each picture (no picture end inside), picture by picture -- picture 0 runs before
any picture end has copied anything on the second-pass stream.  usage: probe_regime.py [--gops G]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gops", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4)
    a = ap.parse_args()
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    W, H, F, G = 1280, 512, a.frames, a.gops
    fb = W * H * 3 // 2
    cfg = encode.config(W, H, qp=32, keyint=1, bframes=0, preset="medium", threads=8)
    enc = encode.BatchEncoder(cfg, G, F)
    d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device="cuda")
    for g in range(G):
        for f in range(F):
            enc.push(g, f, d[f * fb:(f + 1) * fb])
    enc.begin()
    per = enc.advance(0) // F
    half = per // 2
    for pic in range(F):
        enc.sync()
        t = time.perf_counter()
        enc.advance(half)
        enc.sync()
        first = 1e3 * (time.perf_counter() - t) / half
        t = time.perf_counter()
        enc.advance(per - half)
        enc.sync()
        rest = 1e3 * (time.perf_counter() - t) / (per - half)
        print(json.dumps({"picture": pic, "ms_per_step_first_half": round(first, 2), "ms_per_step_second_half_incl_picture_end": round(rest, 2)}), flush=True)


if __name__ == "__main__":
    main()
