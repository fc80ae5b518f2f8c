#!/usr/bin/env python3
"""Times G closed GOPs of ONE clip coded as K independent batch encoders of G/K GOPs each, every encoder driven by its own host thread (its own streams, its own launch
queue): does the device take more launches per second from K threads than from one?  Every bitstream must come out equal (one clip).
usage: probe_par.py --width W --height H --gops G --parts 1,2,4 --frames F"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gops", type=int, default=656)
    ap.add_argument("--parts", default="1,2,4")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    dev = torch.device("cuda", 0)
    W, H, F, G = a.width, a.height, a.frames, a.gops
    fb = W * H * 3 // 2
    cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=a.threads)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    clip = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device=dev, generator=gen)
    torch.cuda.synchronize()
    for K in [int(x) for x in a.parts.replace("+", ",").split(",")]:
        sizes = [G // K + (1 if i < G % K else 0) for i in range(K)]
        encs = []
        for n in sizes:
            e = encode.BatchEncoder(cfg, n, F)
            for g in range(n):
                for f in range(F):
                    e.push(g, f, clip[f * fb:(f + 1) * fb])
            encs.append(e)
        for e in encs:
            e.sync()
        out = [None] * K
        wall = [0.0] * K

        def run(i):
            t = time.perf_counter()
            out[i] = encs[i].encode()
            wall[i] = time.perf_counter() - t

        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        md5 = set(hashlib.md5(b).hexdigest() for o in out for b in o)
        st = [e.stats() for e in encs]
        print(json.dumps({"parts": K, "gops": sizes, "wall_s": round(dt, 2), "frames_per_s": round(G * F / dt, 3), "per_part_wall_s": [round(w, 2) for w in wall],
                          "distinct_bitstreams": len(md5), "md5": sorted(md5)[:2], "host_issue_s": [round(s["step_seconds"], 2) for s in st],
                          "picture_end_s": [round(s["picture_end_seconds"], 2) for s in st]}), flush=True)
        for e in encs:
            e.close()
        del encs, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
