#!/bin/bash
# round 2, final measurements: the bench line as the driver runs it, kernel stats of the step (concurrent and serial), of the intra phase
set -x
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
cut -c1-600 $O/bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/step -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 > $GRAFT_REPO_ROOT/$O/step.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/serial -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 --serial > $GRAFT_REPO_ROOT/$O/serial.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/intra -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 --intra > $GRAFT_REPO_ROOT/$O/intra.log 2>&1 )
tail -1 $O/step.log $O/serial.log; grep "all levels" $O/intra.log
rm -f $O/*/st_kernel_trace.csv
