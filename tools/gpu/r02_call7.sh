#!/bin/bash
# round 2, call 7: PMC passes + kernel stats of the current build, full bench line
set -x
export TMPDIR=/tmp
bash tools/gpu/r02_pmc.sh > gpurun_out/r02pmc.log 2>&1
tail -3 gpurun_out/r02pmc.log
O=gpurun_out/r02c7
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_step -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1 )
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err
cut -c1-1500 $O/bench_full.json
