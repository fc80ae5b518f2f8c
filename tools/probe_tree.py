"""Time of one CTU step of the device-side I-picture tree walk (xeve_hip_mode_analyze_ctu_intra_jobs) against the number of chains in lockstep.
Every chain is a picture of its own (128x128, four CTUs; noise = every node of the tree is visited and decided, smooth = the early termination prunes);
a step codes the same CTU of every picture.  usage: python tools/probe_tree.py [--chains=1,64,256,1024] [--content=noise|smooth]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402


def params(w, h, qp8=32, bd=10, max_cu=32):
    P = lib.TreeParams()
    qp = qp8 + 6 * (bd - 8)
    P.ip.w_scu, P.ip.h_scu, P.ip.slice_type, P.ip.chroma_format_idc, P.ip.bit_depth = w // 4, h // 4, 2, 1, bd
    P.ip.qp[0], P.ip.qp[1], P.ip.qp[2] = qp, qp - 1, qp - 2
    lam = 0.57 * 2.0 ** ((qp8 - 12) / 3.0)
    P.ip.lambda_[0], P.ip.sqrt_lambda0 = lam, lam ** 0.5
    P.ip.dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = 2.0 ** (1 / 3.0), 2.0 ** (2 / 3.0)
    P.ip.lambda_[1], P.ip.lambda_[2] = lam / P.ip.dist_chroma_weight[0], lam / P.ip.dist_chroma_weight[1]
    P.pic_w, P.pic_h, P.log2_ctu, P.max_cu, P.min_cu, P.min_cuwh, P.slice_qp = w, h, 6, max_cu, 4, 4, qp
    return P


def main():
    chains = [1, 64, 256, 1024]
    content = "noise"
    for a in sys.argv[1:]:
        if a.startswith("--chains="):
            chains = [int(v) for v in a.split("=")[1].split(",")]
        if a.startswith("--content="):
            content = a.split("=")[1]
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    w = h = 128
    P = params(w, h)
    g = torch.Generator(device=dev).manual_seed(7)
    for n in chains:
        if content == "noise":
            org = [torch.randint(0, 1024, (n, h, w), device=dev, generator=g, dtype=torch.int16), torch.randint(0, 1024, (n, h // 2, w // 2), device=dev, generator=g, dtype=torch.int16),
                   torch.randint(0, 1024, (n, h // 2, w // 2), device=dev, generator=g, dtype=torch.int16)]
        else:
            yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
            base = (512 + 300 * torch.sin(xx / 19.0) * torch.cos(yy / 23.0))
            luma = (base[None] + torch.randint(-3, 4, (n, h, w), device=dev, generator=g)).clamp(0, 1023).to(torch.int16)
            org = [luma, luma[:, ::2, ::2].contiguous(), luma[:, ::2, ::2].contiguous()]
        mod = [torch.full_like(t, 512) for t in org]
        ms, mc = torch.zeros((n, P.ip.w_scu * P.ip.h_scu), dtype=torch.int32, device=dev), torch.zeros((n, P.ip.w_scu * P.ip.h_scu), dtype=torch.int32, device=dev)
        mi, mt = torch.zeros((n, P.ip.w_scu * P.ip.h_scu), dtype=torch.int8, device=dev), torch.zeros((n, P.ip.w_scu * P.ip.h_scu), dtype=torch.uint8, device=dev)
        st = np.zeros(n, lib.SBAC_DTYPE)
        st["range"], st["code_bits"], st["ctx"] = 16384, 11, 512
        states = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
        need = lib.load().xeve_hip_mode_analyze_ctu_intra_workspace(n, __import__("ctypes").byref(P))
        ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        pe = (org[0][0].numel(), org[1][0].numel(), mod[0][0].numel(), mod[1][0].numel(), ms.shape[1])
        times = []
        for (x, y) in [(0, 0), (64, 0), (0, 64), (64, 64)]:
            jobs = np.zeros(n, lib.CTU_JOB_DTYPE)
            jobs["x"], jobs["y"], jobs["sbac"], jobs["pic"] = x, y, np.arange(n), np.arange(n)
            jt = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, nxt, cost = D.mode_analyze_ctu_intra_jobs([t.data_ptr() for t in org], w, w // 2, [t.data_ptr() for t in mod], w, w // 2, ms, mi, mt, mc, states, P, jt,
                                                           pic_elems=pe, workspace=ws)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            times.append((t1 - t0, t2 - t0))
            states = nxt.clone()
        d = out.cpu().numpy().reshape(-1).view(np.dtype(lib.CTU_DATA_DTYPE))["depth"]
        step = min(t[1] for t in times[1:])
        print("chains %5d  workspace %7.1f MB  step %8.2f ms (host issue %8.2f ms)  -> %9.0f CTUs/s = %7.2f 4K pictures/s (2040 CTUs)   mean depth %.2f" %
              (n, need / 1e6, step * 1e3, min(t[0] for t in times[1:]) * 1e3, n / step, n / step / 2040, float(d.mean())), flush=True)


if __name__ == "__main__":
    main()
