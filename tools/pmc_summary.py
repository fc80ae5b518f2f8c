#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 counter passes: pmc_summary.py OUT.json DIR [DIR ...]  (each DIR holds one *_counter_collection.csv of one --pmc pass).
Kernel names are cut at the first '(' and grouped; the first `skip` launches of every kernel (warm-up call) are dropped."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    dur[(k, os.path.basename(d))].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    res = {}
    for k, cs in sorted(acc.items()):
        res[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
        ds = [x for (kk, _), v in dur.items() if kk == k for x in v]
        res[k]["_duration_ns"] = {"launches": len(ds), "mean": sum(ds) / max(1, len(ds))}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
