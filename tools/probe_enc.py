#!/usr/bin/env python3
"""Times the batch encoder in chunks of lockstep steps (device fence after every chunk): where a picture's time goes -- the steps of the I picture, of the inter
picture, the picture ends.  usage: probe_enc.py --width W --height H --gops G --threads T --frames F --chunk N [--content noise|moving] [--preset P]"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gops", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=25)
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many lockstep steps (0: the whole job)")
    ap.add_argument("--content", default="noise")
    ap.add_argument("--preset", default="medium", help="fast | medium | slow | placebo (slow and placebo run on the fused walk at any width)")
    ap.add_argument("--same-clip", action="store_true", help="every GOP codes the SAME frames: all bitstreams must come out equal -- a lockstep chain that goes wrong anywhere in the batch shows as a second md5")
    ap.add_argument("--prof-after", type=int, default=-1, help="switch the walk's in-kernel stage profile on once this many steps are done (e.g. a picture's steps: the inter picture alone)")
    a = ap.parse_args()
    import torch

    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    dev = torch.device("cuda", 0)
    W, H, F, G = a.width, a.height, a.frames, a.gops
    fb = W * H * 3 // 2
    cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset=a.preset, threads=a.threads)
    t0 = time.perf_counter()
    enc = encode.BatchEncoder(cfg, G, F)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    same = None
    for g in range(G):
        if a.same_clip and same is not None:
            d = same
        elif a.content == "noise":
            d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device=dev, generator=gen)
        else:  # a drifting gradient + 3 bits of noise (SURVEY.md 8d's structured input)
            parts = []
            for f in range(F):
                yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
                parts.append((((xx + 3 * f + g) * 2 + (yy + f) + torch.randint(0, 8, (H, W), device=dev, generator=gen)) & 255).to(torch.uint8).reshape(-1))
                yy, xx = torch.meshgrid(torch.arange(H // 2, device=dev), torch.arange(W // 2, device=dev), indexing="ij")
                parts.append((((xx + f) * 3 + yy + torch.randint(0, 4, (H // 2, W // 2), device=dev, generator=gen)) & 255).to(torch.uint8).reshape(-1))
                parts.append(((xx + (yy + 2 * f) * 2 + torch.randint(0, 4, (H // 2, W // 2), device=dev, generator=gen)) & 255).to(torch.uint8).reshape(-1))
            d = torch.cat(parts)
        same = d
        for f in range(F):
            enc.push(g, f, d[f * fb:(f + 1) * fb])
    print(json.dumps({"setup_s": round(time.perf_counter() - t0, 2)}), flush=True)
    enc.begin()
    total = enc.advance(0)
    per_pic = total // F
    done, left = 0, total
    while left > 0 and (a.max_steps == 0 or done < a.max_steps):
        n = min(a.chunk, per_pic - done % per_pic)  # (a chunk never crosses a picture's end: its last step carries the picture end)
        if a.prof_after >= 0 and done >= a.prof_after and not os.environ.get("XEVE_HIP_WALK_PROF"):
            from xeve_amd import lib
            lib.load().xeve_hip_walk_prof_enable(1)
            os.environ["XEVE_HIP_WALK_PROF"] = "runtime"
        t = time.perf_counter()
        st0 = enc.stats()
        left = enc.advance(n)
        enc.sync()
        dt = time.perf_counter() - t
        st1 = enc.stats()
        done += n
        print(json.dumps({"steps": [done - n, done], "picture": (done - 1) // per_pic, "wall_ms_per_step": round(1e3 * dt / n, 2),
                          "host_issue_ms_per_step": round(1e3 * (st1["step_seconds"] - st0["step_seconds"]) / n, 2),
                          "picture_end_s": round(st1["picture_end_seconds"] - st0["picture_end_seconds"], 2)}), flush=True)
    if os.environ.get("XEVE_HIP_WALK_PROF"):
        from tools.probe_walk import prof
        rows = prof()
        tot = sum(r[1] for r in rows) or 1
        print("walk profile of team 0 over the whole run: %.0f cycles" % tot)
        for name, cyc, marks in sorted(rows, key=lambda r: -r[1]):
            if marks:
                print("  %-12s %6.2f %%  %12d cycles  %7d marks  %8.0f cycles/mark" % (name, 100.0 * cyc / tot, cyc, marks, cyc / marks))
    print(json.dumps({"total_steps": total, "per_picture": per_pic, "chains": G * min(a.threads, (H + 63) // 64), "stats": enc.stats(),
                      "bytes": [len(s) for s in enc.bitstreams()][:4] if left == 0 else None,
                      "md5_of_all": hashlib.md5(b"".join(enc.bitstreams())).hexdigest() if left == 0 else None,
                      "distinct_bitstreams": (lambda m: {"count": len(set(m)), "gops_that_differ_from_gop_0": [i for i, x in enumerate(m) if x != m[0]][:40]})([hashlib.md5(b).hexdigest() for b in enc.bitstreams()]) if left == 0 and a.same_clip else None}), flush=True)
    enc.close()


if __name__ == "__main__":
    main()
