#!/bin/bash
mkdir -p gpurun_out/tree_prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/tree_prof -o tree1 -- python /root/repo/tools/probe_tree.py --chains=1 > /root/repo/gpurun_out/r02_tree_prof.log 2>&1
cd /root/repo
ls -R gpurun_out/tree_prof | head -20
f=$(find gpurun_out/tree_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02_tree1_kernel_stats.csv
head -50 gpurun_out/r02_tree1_kernel_stats.csv | cut -c1-200
find gpurun_out/tree_prof -name "*kernel_trace.csv" -delete
