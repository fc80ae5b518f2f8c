#!/usr/bin/env python3
"""Timing of xeve_hip_pinter_analyze_cu_jobs: the whole inter analysis of every CU of a picture at one size (B slice, 2 reference pictures per list)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _inter_cases import make_inter_jobs, make_inter_params, make_inter_picture  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402
from _rdo_cases import states  # noqa: E402
from test_hip_inter import hip_params  # noqa: E402

import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
(w, h), bd, nref = ((3840, 2176) if "--4k" in sys.argv else (1920, 1088)), 10, (1 if "--medium" in sys.argv else 2)  # me_ref_num = 1 in fast / medium / slow
st_type = 1 if "--p" in sys.argv else 0
r = np.random.default_rng(1)
refs, org = make_inter_picture(r, w, h, bd, nref, 1, st_type)
st = states(r, 64)
dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
dorg = [torch.from_numpy(x).to(dev) for x in org]
org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
dst = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
total = 0.0
levels = []
for lw in (3, 4, 5, 6):
    c = 1 << lw
    n = (w // c) * (h // c)
    P = make_inter_params(r, lw, w, h, bd, nref, 1, st_type, refs, 0.0, max_cand=3)
    if "--medium" in sys.argv:  # preset medium: me_range 64, half-pel search with 4 positions, no quarter-pel stage, 3 merge candidates (xeve_enc.c:2455-2471)
        P.me.max_search_range, P.spel.hpel_cnt, P.spel.qpel_cnt = 64, 4, 0
    hp = hip_params(P)
    jobs = make_inter_jobs(r, n, w, h, c, len(st), refs, st_type)
    jobs["x"], jobs["y"] = (np.arange(n) % (w // c)) * c, (np.arange(n) // (w // c)) * c
    dj = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    need = lib.load().xeve_hip_pinter_analyze_cu_workspace(n, len(st), ctypes.byref(hp), refs["s_l"], refs["s_c"])
    levels.append((c, n, hp, dj, torch.empty(int(need), dtype=torch.uint8, device=dev), torch.cuda.Stream()))
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = D.pinter_analyze_cu_jobs(org_ptrs, refs["s_l"], refs["s_c"], tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=levels[-1][4])[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    m = res.cpu().numpy().reshape(-1).view(np.dtype(lib.INTER_RESULT_DTYPE))["best_idx"]
    total += dt
    print("%dx%d %s CU %2d: %6d CUs: %.2f ms (workspace %.0f MB)  modes L0/L1/BI/SKIP/DIR %s" % (w, h, "P" if st_type else "B", c, n, dt * 1e3, need / 1e6, np.bincount(m, minlength=5)),
          flush=True)
print("all four levels, one after the other: %.2f ms" % (total * 1e3))
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (c, n, hp, dj, ws, strm) in reversed(levels):  # the latency-bound large-CU levels first
        with torch.cuda.stream(strm):
            D.pinter_analyze_cu_jobs(org_ptrs, refs["s_l"], refs["s_c"], tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("all four levels, one stream per level: %.2f ms" % (dt * 1e3))

# the same four calls recorded into one HIP graph (every launch of the composite goes to the caller's stream; no host round trip inside)
try:
    outs = {}
    pre = {c: [torch.empty_like(t) for t in D.pinter_analyze_cu_jobs(org_ptrs, refs["s_l"], refs["s_c"], tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=ws)]
           for (c, n, hp, dj, ws, strm) in levels}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for (c, n, hp, dj, ws, strm) in reversed(levels):
            strm.wait_stream(main)
            with torch.cuda.stream(strm):
                outs[c] = D.pinter_analyze_cu_jobs(org_ptrs, refs["s_l"], refs["s_c"], tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=ws)
        for (c, n, hp, dj, ws, strm) in levels:
            main.wait_stream(strm)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("all four levels, one HIP graph (a stream per level inside): %.2f ms" % (dt * 1e3))
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", repr(e)[:300])
