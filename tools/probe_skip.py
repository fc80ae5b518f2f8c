#!/usr/bin/env python3
"""Timing of xeve_hip_analyze_skip_jobs: every CU of a picture at one size, B slice, 4 candidates (16 pairs)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _mc_cases import refpic_table  # noqa: E402
from _rdo_cases import make_params, make_picture, make_skip_jobs, states  # noqa: E402

import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
(w, h), bd, nref = ((3840, 2176) if "--4k" in sys.argv else (1920, 1088)), 10, 2
r = np.random.default_rng(1)
refs, org = make_picture(r, w, h, bd, nref, 1)
st = states(r, 64)
dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
dorg = [torch.from_numpy(x).to(dev) for x in org]
org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
dst = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
for lw in (3, 4, 5, 6):
    for ncand, want in ((4, True), (3, False)):
        c = 1 << lw
        n = (w // c) * (h // c)
        p = make_params(r, lw, lw, w, h, bd, nref, 1, 0)
        hp = lib.RdoParams.from_buffer_copy(bytes(p))
        jobs = make_skip_jobs(r, n, w, h, c, c, len(st), ncand)
        jobs["x"], jobs["y"] = (np.arange(n) % (w // c)) * c, (np.arange(n) // (w // c)) * c
        dj = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        need = lib.load().xeve_hip_analyze_skip_workspace(n, ctypes.byref(hp), ncand)
        ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D.analyze_skip_jobs(org_ptrs, refs["s_l"], refs["s_c"], tab, refs["s_l"], refs["s_c"], dst, hp, dj, max_cand=ncand, want_state=want, workspace=ws)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%dx%d CU %2d: %6d CUs x %2d pairs, state %d: %.2f ms (workspace %.0f MB)" % (w, h, c, n, ncand * ncand, want, dt * 1e3, need / 1e6), flush=True)
