#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c15
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/l64 -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 --sizes=64 > $GRAFT_REPO_ROOT/$O/l64.log 2>&1 )
tail -1 $O/l64.log
