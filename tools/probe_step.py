#!/usr/bin/env python3
"""One bench.py step under a profiler: HotPathPass.inter() (the whole inter analysis of every CU of every level of one picture), `reps` times after one
warm-up call.  Usage: probe_step.py [reps] [--1080p] [--structured] [--serial] [--mfma] [--intra] [--graph]
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step -o st -- python tools/probe_step.py 3
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python tools/probe_step.py 1
--serial runs the four levels one after the other on one stream (per-kernel counters are cleaner without overlap);
--mfma only runs the 32x32 / 64x64 fused residual chain (k_rdo_mfma) on every block of the picture, for the MFMA counters."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd.workload import HotPathPass  # noqa: E402

args = [x for x in sys.argv[1:] if not x.startswith("--")]
reps = int(args[0]) if args else 3
w, h = (1920, 1080) if "--1080p" in sys.argv else (3840, 2160)
xeve_amd.init(0)
dev = torch.device("cuda:0")
sizes = tuple(int(v) for v in [x for x in sys.argv if x.startswith("--sizes=")][0].split("=")[1].split(",")) if any(x.startswith("--sizes=") for x in sys.argv) else (8, 16, 32, 64)
wl = HotPathPass(w, h, dev, seed=4, content="structured" if "--structured" in sys.argv else "iid", sizes=sizes)  # --sizes=64,32: only those CU levels
if "--mfma" in sys.argv:
    wl.run(only="D1")  # predictions
    for _ in range(reps + 1):
        for S in (32, 64):
            lv = wl.lv[S]
            l2 = S.bit_length() - 1
            D.residual_rdo(wl.org[0], wl.s_l, lv["pred_l"][0], S, lv["dense_jobs"], l2, l2, wl.bd, wl.qp, False, True, lv["coef"][0], lv["rec"][0], wl.s_l, lv["nnz"][0],
                           lv["ssd2"][0])
    torch.cuda.synchronize()
    sys.exit(0)
if "--intra" in sys.argv:  # phase I: the intra analysis of every CU of every level (64 .. 4)
    wl.intra()
    torch.cuda.synchronize()
    for sizes in (None,) + tuple((S,) for S in wl.INTRA_SIZES):
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.intra(sizes)
        torch.cuda.synchronize()
        print("%dx%d %s intra %s: %.2f ms per step (%d reps)" % (w, h, wl.content, "all levels" if sizes is None else "%dx%d" % (sizes[0], sizes[0]),
                                                                  1e3 * (time.perf_counter() - t0) / reps, reps), flush=True)
    sys.exit(0)
if "--serial" in sys.argv:
    one = torch.cuda.current_stream()
    wl._side = {S: one for S in wl.sizes}
npic = int([x for x in sys.argv if x.startswith("--pictures=")][0].split("=")[1]) if any(x.startswith("--pictures=") for x in sys.argv) else 1
wls = [wl] + [HotPathPass(w, h, dev, seed=4 + k, content=wl.content) for k in range(1, npic)]  # --pictures=N: N independent pictures in flight, each on its own streams
for p in wls:
    p.inter()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    for p in wls:
        p.inter()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("%dx%d %s: %.2f ms per step (%d reps)%s" % (w, h, wl.content, 1e3 * dt, reps, "" if npic == 1 else ", %d pictures per step = %.2f pictures/s" % (npic, npic / dt)), flush=True)
if "--graph" in sys.argv:  # the same step captured into one HIP graph (all four level streams fork from and join the capture stream) and replayed
    cs = torch.cuda.Stream(device=dev)
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        wl.inter()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cs):
            wl.inter()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    print("%dx%d %s: %.2f ms per step replayed from ONE HIP graph (%d reps)" % (w, h, wl.content, 1e3 * (time.perf_counter() - t0) / reps, reps), flush=True)
