#!/usr/bin/env python3
"""Developer probe: xeve_hip_residue_rdo_jobs for every CU of a 1920x1080 picture (one candidate per CU, vectors near the true
motion), per CU size; and the oracle's single-thread time per candidate for scale."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xeve_amd  # noqa: E402
from _libs import RDO_RESULT_DTYPE, SBAC_DTYPE, oracle_rdo, ptr  # noqa: E402
from _mc_cases import refpic_table  # noqa: E402
from _rdo_cases import make_jobs, make_params, make_picture, states  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
W, H, bd, nref = 1920, 1080, 10, 2
r = np.random.default_rng(3)
refs, org = make_picture(r, W, H, bd, nref, 1)
st = states(r, 16)
dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
host_tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
dorg = [torch.from_numpy(x).to(dev) for x in org]
org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
horg = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
dst = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
O = oracle_rdo()
for lw in (3, 4, 5, 6):
    S = 1 << lw
    p = make_params(r, lw, lw, W, H, bd, nref, 1, 0)
    p.qp[0], p.qp[1], p.qp[2] = 44, 44, 44  # qp 32 + 12
    nx, ny = W // S, H // S
    jobs = make_jobs(r, nx * ny, W, H, S, S, nref, len(st), 0)
    ys, xs = np.meshgrid(np.arange(ny) * S, np.arange(nx) * S, indexing="ij")
    jobs["x"], jobs["y"] = xs.ravel(), ys.ravel()
    jobs["mv"] = r.integers(-2, 3, size=(len(jobs), 2, 2))
    jobs["refi"][:, 0], jobs["dir_flag"] = 0, 0
    hp = lib.RdoParams.from_buffer_copy(bytes(p))
    dj = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    need = lib.load().xeve_hip_residue_rdo_workspace(len(jobs), len(st), __import__("ctypes").byref(hp), refs["s_l"], refs["s_c"])
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
    for _ in range(2):
        res, coef, best = D.residue_rdo_jobs(org_ptrs, refs["s_l"], refs["s_c"], dev_tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        res, coef, best = D.residue_rdo_jobs(org_ptrs, refs["s_l"], refs["s_c"], dev_tab, refs["s_l"], refs["s_c"], dst, hp, dj, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    rr = res.cpu().numpy().reshape(-1).view(RDO_RESULT_DTYPE)
    # oracle, single thread, on a sample
    nsamp = min(200, len(jobs))
    t0 = time.perf_counter()
    for i in range(nsamp):
        er, eb = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
        ec = [np.zeros(S * S, np.int16), np.zeros(S * S // 4, np.int16), np.zeros(S * S // 4, np.int16)]
        O.xo_residue_rdo(ptr(horg), refs["s_l"], refs["s_c"], ptr(host_tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]), ptr(ec[1]),
                         ptr(ec[2]), ptr(eb))
        assert er["cost"][0] == rr["cost"][i]
    t1 = time.perf_counter()
    print("%2dx%-2d candidates %6d  GPU %.3f ms per picture (%.2f us each)   oracle 1 thread %.1f us each   coded %.1f%%  workspace %.0f MB"
          % (S, S, len(jobs), e0.elapsed_time(e1) / 3, 1e3 * e0.elapsed_time(e1) / 3 / len(jobs), 1e6 * (t1 - t0) / nsamp,
             100.0 * np.count_nonzero(rr["nnz"].any(axis=1)) / len(jobs), need / 1e6), flush=True)
