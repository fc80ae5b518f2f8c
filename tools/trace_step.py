#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace CSV of an encode on the composed walk and writes what ONE lockstep step is made of: the launch sequence of a step from the
middle of the last picture (kernel, grid, duration, gap to the previous kernel's end), the per-kernel totals of that step and the sum of the gaps.
A step starts with the walk's memset (`__amd_rocclr_fillBufferAligned` followed by k_tree_ops).
usage: trace_step.py kernel_trace.csv out_prefix [step_from_end]"""
import csv
import json
import re
import sys
from collections import OrderedDict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", n)
    return m.group(1) if m else n[:60]


def main():
    path, out = sys.argv[1], sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                     int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), r.get("Queue_Id", "?")))
    rows.sort()
    # the walk's queue: the one k_tree_ops runs on
    q = next(r[5] for r in rows if r[2].startswith("k_tree_ops"))
    main_q = [r for r in rows if r[5] == q]
    starts = [i for i in range(len(main_q) - 1) if "fillBuffer" in main_q[i][2] and main_q[i + 1][2].startswith("k_tree_ops")]
    print("launches", len(rows), "on the walk's queue", len(main_q), "steps", len(starts))
    if len(starts) < back + 1:
        back = len(starts) - 1
    a, b = starts[-back - 1], starts[-back]
    t0, t1 = main_q[a][0], main_q[b][0]
    # every queue's launches inside the step's window: the walk's own queue and (round 6) the side stream's
    step = [r for r in rows if t0 <= r[0] < t1]
    queues = sorted(set(r[5] for r in step), key=lambda x: -sum(1 for r in step if r[5] == x))
    busy = {}
    for qq in queues:
        iv = sorted((r[0], r[1]) for r in step if r[5] == qq)
        tot, end = 0, t0
        for s_, e_ in iv:
            tot += max(0, e_ - max(s_, end))
            end = max(end, e_)
        busy[qq] = {"launches": len(iv), "busy_ms": tot / 1e6, "first_ms": (iv[0][0] - t0) / 1e6, "last_end_ms": (iv[-1][1] - t0) / 1e6}
    # time covered by at least one kernel of any queue, and by at least two
    evs = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    cov1 = cov2 = 0
    depth, last = 0, t0
    for t, d in evs:
        if depth >= 1:
            cov1 += t - last
        if depth >= 2:
            cov2 += t - last
        depth += d
        last = t
    per, gaps, glue, seq = OrderedDict(), 0, 0, []
    prev_end = {}
    for s, e, n, g, wg, qq in step:
        gap = s - prev_end.get(qq, s)
        prev_end[qq] = max(prev_end.get(qq, e), e)
        gaps += max(0, gap)
        d = per.setdefault(n, [0, 0, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = max(d[2], e - s)
        if e - s < 8000:
            glue += 1
        seq.append((n, g, wg, e - s, gap, queues.index(qq)))
    summary = {"launches_in_step": len(step), "step_wall_ms": (t1 - t0) / 1e6, "kernel_ms": sum(v[1] for v in per.values()) / 1e6, "gap_ms": gaps / 1e6,
               "launches_under_8us": glue, "steps_in_trace": len(starts), "queues": busy, "covered_by_a_kernel_ms": cov1 / 1e6, "covered_by_two_ms": cov2 / 1e6}
    print(json.dumps(summary))
    with open(out + "_kernels.csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us,max_us\n")
        for n, (c, t, m) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.1f,%.2f,%.1f\n" % (n.replace(",", ";"), c, t / 1e3, t / 1e3 / c, m / 1e3))
    # one node's worth of the sequence is enough to read the chain: write the whole step, it is ~ 15 000 lines of < 60 bytes
    with open(out + "_sequence.csv", "w") as f:
        f.write("kernel,grid,wg,dur_ns,gap_ns,queue\n")
        for n, g, wg, d, gap, qi in seq:
            f.write("%s,%d,%d,%d,%d,%d\n" % (n.replace(",", ";"), g, wg, d, gap, qi))
    json.dump(summary, open(out + "_summary.json", "w"))


if __name__ == "__main__":
    main()
