#!/bin/bash
# round 2, call 13: stream priorities for the big-CU levels; several pictures in flight
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c13
mkdir -p $O
for prio in 0 1 2; do echo "prio=$prio: $(XEVE_HIP_LEVEL_PRIO=$prio python tools/probe_step.py 5 2>&1 | tail -1)"; done > $O/prio.log 2>&1
cat $O/prio.log
for q in 4 8; do for n in 1 2 3; do echo "queues=$q pictures=$n: $(GPU_MAX_HW_QUEUES=$q python tools/probe_step.py 4 --pictures=$n 2>&1 | tail -1)"; done; done > $O/pictures.log 2>&1
cat $O/pictures.log
