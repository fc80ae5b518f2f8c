#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c17
mkdir -p $O
for sz in 8 32; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/l$sz -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 --sizes=$sz > $GRAFT_REPO_ROOT/$O/l$sz.log 2>&1 )
tail -1 $O/l$sz.log
done
