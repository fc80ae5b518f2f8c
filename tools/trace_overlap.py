#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace CSV and reports how much of the second writer pass (k_enc_rewrite) ran while other kernels ran: trace_overlap.py enc_kernel_trace.csv"""
import csv
import sys

rw, other = [], []
for r in csv.DictReader(open(sys.argv[1])):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    (rw if "k_enc_rewrite" in r["Kernel_Name"] else other).append((a, b, r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
other.sort()
print("rewrite launches", len(rw), "other kernels", len(other))
import bisect
starts = [o[0] for o in other]
tot = ov = n_in = 0
for a, b, *_ in rw:
    tot += b - a
    i = bisect.bisect_left(starts, a)
    j = bisect.bisect_right(starts, b)
    n_in += j - i
    for o in other[max(0, i - 1):j]:
        ov += max(0, min(b, o[1]) - max(a, o[0]))
print("rewrite total ms", tot / 1e6, "other-kernel time inside rewrite intervals ms", ov / 1e6, "other kernels started inside", n_in)
qs = {}
for a, b, name, q, s in rw[:3] + other[:3] + other[-3:]:
    print(name, "queue", q, "stream", s, "dur us", (b - a) / 1e3)
if rw:
    a0, b0 = rw[0][0], rw[-1][1]
    inside = [o for o in other if o[0] >= a0 and o[0] <= b0]
    print("first rewrite window ms", (b0 - a0) / 1e6, "other kernels started in the window of the FIRST picture's pass:", len([o for o in other if rw[0][0] <= o[0] <= rw[len(rw) // 2 - 1][1]]))
