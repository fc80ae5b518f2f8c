#!/bin/bash
# round 2, call 12: host issue of the four levels -- threads x hardware queues x the one-round RDO decision
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c12
mkdir -p $O
for q in 4 8; do for t in 0 1; do for spec in 256 3000 9000 40000; do
  echo "queues=$q threads=$t spec=$spec: $(GPU_MAX_HW_QUEUES=$q XEVE_HIP_LEVEL_THREADS=$t XEVE_HIP_RDO_SPEC=$spec python tools/probe_step.py 5 2>&1 | tail -1)"
done; done; done > $O/matrix.log 2>&1
cat $O/matrix.log
echo "structured: $(python tools/probe_step.py 5 --structured 2>&1 | tail -1)" >> $O/matrix.log
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 2 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
tail -2 $O/matrix.log
timeout 300 python -m pytest tests/test_workload.py -x -q -m gpu -k inter 2>&1 | tail -3
