#!/usr/bin/env python3
"""Developer probe: time xeve_hip_cu_bits_jobs for every CU of a 3840x2160 picture, per CU size, dense vs sparse levels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
W, H = 3840, 2160
r = np.random.default_rng(1)
for kind in ("dense", "sparse"):
    for S in (8, 16, 32, 64):
        lw = S.bit_length() - 1
        n = (W // S) * (H // S)
        ny, nc = S * S, S * S // 4
        per = ny + 2 * nc
        if kind == "dense":
            coef = r.integers(-12, 13, size=n * per).astype(np.int16)
        else:
            coef = np.where(r.random(n * per) < 0.03, r.integers(-2, 3, size=n * per), 0).astype(np.int16)
        jobs = np.zeros(n, lib.CU_BITS_JOB_DTYPE)
        base = np.arange(n) * per
        jobs["coef_off"][:, 0], jobs["coef_off"][:, 1], jobs["coef_off"][:, 2] = base, base + ny, base + ny + nc
        c3 = coef.reshape(n, per)
        jobs["nnz"][:, 0] = np.count_nonzero(c3[:, :ny], axis=1)
        jobs["nnz"][:, 1] = np.count_nonzero(c3[:, ny:ny + nc], axis=1)
        jobs["nnz"][:, 2] = np.count_nonzero(c3[:, ny + nc:], axis=1)
        jobs["refi"][:, 0], jobs["refi"][:, 1] = 0, -1
        jobs["mvd"][:, 0] = r.integers(-40, 41, size=(n, 2))
        st = np.zeros(1, lib.SBAC_DTYPE)
        st["range"], st["ctx"] = 16384, 512
        p = lib.CuBitsParams()
        p.log2_cuw = p.log2_cuh = lw
        p.num_refp[0] = p.num_refp[1] = 2
        p.chroma_format_idc = 1
        dc, dj, ds = torch.from_numpy(coef).to(dev), torch.from_numpy(jobs.view(np.uint8)).to(dev), torch.from_numpy(st.view(np.uint8)).to(dev)
        ws = torch.empty(int(lib.load().xeve_hip_cu_bits_workspace(n, dc.numel())), dtype=torch.uint8, device=dev)
        bits = torch.empty(n, dtype=torch.int32, device=dev)
        for want in (False,):
            for _ in range(2):
                D.cu_bits_jobs(dc, ds, dj, p, want_state=want, workspace=ws, bits=bits)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                D.cu_bits_jobs(dc, ds, dj, p, want_state=want, workspace=ws, bits=bits)
            e1.record()
            torch.cuda.synchronize()
            tb = int(bits.sum().item())
            print("%-6s %2dx%-2d jobs %6d  %.3f ms/picture  total bits %d (%.1f bits/coef)  %.1f Mbin-ish/s"
                  % (kind, S, S, n, e0.elapsed_time(e1) / 5, tb, tb / (n * per), tb / (e0.elapsed_time(e1) / 5) / 1e3), flush=True)
