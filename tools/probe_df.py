#!/usr/bin/env python3
"""Developer probe: time xeve_hip_deblock + xeve_hip_picbuf_expand on a 3840x2160 4:2:0 picture with a random quad-tree."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xeve_amd  # noqa: E402
from _df_cases import make_case, origin  # noqa: E402
from xeve_amd import device as D  # noqa: E402
from xeve_amd import lib  # noqa: E402

xeve_amd.init(0)
dev = torch.device("cuda:0")
for min_cu in (8, 4):
    c = make_case(np.random.default_rng(1), 3840, 2160, 10, 1, min_cu)
    planes = [torch.from_numpy(p).to(dev) for p in c["planes"]]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    maps = [up(c[k]) for k in ("map_scu", "map_cu_mode", "refi", "mv")]
    p = lib.DeblockParams.from_buffer_copy(bytes(c["p"]))
    org = [origin(c, k) for k in range(3)]
    for name, fn in (("deblock", lambda: D.deblock(planes, org, c["s_l"], c["s_c"], *maps, p)),
                     ("picbuf_expand(16)", lambda: D.picbuf_expand(planes, org, c["s_l"], c["s_c"], 3840, 2160, 1920, 1080, 16, 16, 1))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("min_cu %d  %-18s %.3f ms per 4K picture" % (min_cu, name, e0.elapsed_time(e1) / 10), flush=True)
