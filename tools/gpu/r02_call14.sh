#!/bin/bash
# round 2, call 14: each CU level alone, with the 4-round and the one-round RDO decision
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c14
mkdir -p $O
for sz in 64 32 16 8 64,32 16,8; do for spec in 256 100000; do
  echo "sizes=$sz spec=$spec: $(XEVE_HIP_RDO_SPEC=$spec python tools/probe_step.py 5 --sizes=$sz 2>&1 | tail -1)"
done; done > $O/levels.log 2>&1
cat $O/levels.log
