#!/usr/bin/env python3
"""Developer probe: phase G (pinter_residue_rdo per CU) of the structured 4K picture, per level."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xeve_amd  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd.workload import HotPathPass  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
wl = HotPathPass(3840, 2160, dev, seed=5, content=sys.argv[1] if len(sys.argv) > 1 else "structured")
wl.rdo()
torch.cuda.synchronize()
for S in wl.sizes:
    r = wl.lv[S]["rdo"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        D.residue_rdo_jobs(r["org"], wl.s_l, wl.s_c, r["refp"], wl.s_l, wl.s_c, r["state"], r["params"], r["jobs"], workspace=r["ws"])
    e1.record()
    torch.cuda.synchronize()
    print("%2dx%-2d %6d candidates  %.3f ms" % (S, S, wl.lv[S]["n"], e0.elapsed_time(e1) / 3), flush=True)
