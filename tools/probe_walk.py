"""Per-stage cycle profile of the fused CTU walk (xeve_amd/csrc/walk.h) from its in-kernel marks: XEVE_HIP_WALK_PROF=1 makes thread 0 of team 0 add the cycles between
two stage marks to the stage's class.  usage: XEVE_HIP_WALK_PROF=1 [XEVE_HIP_WALK_C=8] python tools/probe_walk.py [--chains=64] [--content=noise]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xeve_amd  # noqa: E402
from xeve_amd import lib  # noqa: E402
from xeve_amd.workload import CtuWalkIntra  # noqa: E402

NAMES = ["clear", "enter", "leaf", "child_done", "exit", "root", "mid", "i_setup", "i_nbr", "i_pred", "i_satd", "i_list", "i_bits", "i_pick", "i_cpred", "i_final", "b_diff",
         "b_t0", "b_t1", "b_rdoq", "b_dq", "b_t2", "b_t3", "b_rec", "e_cand", "e_skip", "e_me", "e_spel", "e_mc", "e_bits", "e_glue", "e_final", "m_bits", "m_sad", "m_sel", "q_a", "q_b"]


def prof():
    L = lib.load()
    n = 2 * len(NAMES)
    buf = (C.c_uint64 * n)()
    got = L.xeve_hip_walk_prof(buf, n)
    return [(NAMES[i], buf[i], buf[got + i]) for i in range(got)] if got else []


def main():
    chains, content = 64, "noise"
    for a in sys.argv[1:]:
        if a.startswith("--chains="):
            chains = int(a.split("=")[1])
        if a.startswith("--content="):
            content = a.split("=")[1]
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wk = CtuWalkIntra(chains, dev, content)
    for _ in range(2):
        wk.step()
    torch.cuda.synchronize()
    prof()
    t0 = time.perf_counter()
    wk.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = prof()
    tot = sum(r[1] for r in rows) or 1
    print("chains %d  step %.2f ms  team 0: %.0f cycles in %d marks" % (chains, dt * 1e3, tot, sum(r[2] for r in rows)))
    for name, cyc, marks in sorted(rows, key=lambda r: -r[1]):
        if marks:
            print("  %-12s %6.2f %%  %10d cycles  %6d marks  %8.0f cycles/mark" % (name, 100.0 * cyc / tot, cyc, marks, cyc / marks))


if __name__ == "__main__":
    main()
