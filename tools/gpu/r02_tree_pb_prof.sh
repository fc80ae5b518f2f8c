#!/bin/bash
# kernel trace of the P / B tree walk: one B-picture case of tests/test_hip_tree.py (two 64x64 CTUs), one chain
mkdir -p gpurun_out/tree_pb
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tree_pb -o pb -- python -m pytest $GRAFT_REPO_ROOT/tests/test_hip_tree.py -q -k 4102 > $GRAFT_REPO_ROOT/gpurun_out/r02_tree_pb_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tree_pb -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02_tree_pb_kernel_stats.csv
find gpurun_out/tree_pb -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_tree_pb_kernel_stats.csv')))
tot=sum(int(r['Calls']) for r in rows); t=sum(float(r['TotalDurationNs']) for r in rows)
print("launches",tot,"busy ms",t/1e6)
for r in rows[:14]: print(r['Calls'], r['Percentage'], r['Name'][:60])
PY
tail -2 gpurun_out/r02_tree_pb_prof.log
