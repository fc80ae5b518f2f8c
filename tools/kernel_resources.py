#!/usr/bin/env python3
"""Static resources of every kernel of the library (no GPU needed): compiles each .hip to gfx950 assembly and reads the kernel descriptors' metadata -- VGPRs, SGPRs,
LDS bytes, scratch bytes per work-item, instructions -- into a CSV.  usage: kernel_resources.py out.csv"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "xeve_amd", "csrc")
rows = []
with tempfile.TemporaryDirectory() as d:
    for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        out = os.path.join(d, os.path.basename(f) + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, f],
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
        sizes = {}  # instructions per kernel body: lines between the symbol and s_endpgm that start with a tab and an opcode
        for m in re.finditer(r"^(\S+):.*?; @\1\n(.*?)\ts_endpgm", txt, re.S | re.M):
            sizes[m.group(1)] = sum(1 for line in m.group(2).split("\n") if line.startswith("\t") and not line.strip().startswith((";", ".")))
        for m in re.finditer(r"  - \.agpr_count:.*?\n(.*?)(?=\n  - \.agpr_count|\namdhsa\.target)", txt, re.S):
            blk = m.group(0)
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, ""])[1]
            name = g("name")
            try:
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            except Exception:
                dem = name
            rows.append((os.path.basename(f), re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", ""))[:90], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size"), g("max_flat_workgroup_size"), sizes.get(name, "")))
with open(sys.argv[1], "w") as o:
    o.write("file,kernel,vgprs,agprs,sgprs,lds_bytes,scratch_bytes_per_lane,max_workgroup,instructions\n")
    for r in rows:
        o.write(",".join('"%s"' % x if "," in str(x) else str(x) for x in r) + "\n")
print(len(rows), "kernels")
