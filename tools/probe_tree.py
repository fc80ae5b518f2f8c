"""Time of one CTU step of the device-side I-picture tree walk (xeve_hip_mode_analyze_ctu_jobs) against the number of chains in lockstep (xeve_amd/workload.py
CtuWalkIntra: every chain is a 128x128 picture of its own; a step decides the same CTU of every picture).
usage: [XEVE_HIP_TREE_GRAPH=1] python tools/probe_tree.py [--chains=1,64,256,1024] [--content=noise|smooth|texture] [--write]
(with the graph switch on, steps 5 .. 7 are replays and steps 2 .. 3 launch-by-launch; off, all are launch-by-launch)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xeve_amd  # noqa: E402
from xeve_amd.workload import CtuWalkIntra  # noqa: E402


def main():
    chains = [1, 64, 256, 1024]
    content = "noise"
    write = "--write" in sys.argv  # every decided CTU also written on the device (xeve_hip_eco_ctu_jobs): the chains carry the writer's coder state
    for a in sys.argv[1:]:
        if a.startswith("--chains="):
            chains = [int(v) for v in a.split("=")[1].split(",")]
        if a.startswith("--content="):
            content = a.split("=")[1]
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    for n in chains:
        wk = CtuWalkIntra(n, dev, content, write=write)
        times = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wk.step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            times.append((t1 - t0, time.perf_counter() - t0))
        step, eager = min(t[1] for t in times[4:]), min(t[1] for t in times[1:3])
        print("chains %5d  workspace %7.1f MB  step %8.2f ms (host %6.2f ms) [steps 5-7], %8.2f ms (host %6.2f ms) [steps 2-3]  -> %9.0f CTUs/s = %7.2f 4K "
              "pictures/s (2040 CTUs)   mean depth %.2f" % (n, wk.need / 1e6, step * 1e3, min(t[0] for t in times[4:]) * 1e3, eager * 1e3, min(t[0] for t in times[1:3]) * 1e3,
                                                           n / step, n / step / 2040, wk.mean_depth()) +
              ("   [decided AND written: %.0f bytes of slice data per CTU]" % float(wk.bytes[1].float().mean()) if write else ""), flush=True)
        del wk
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
