#!/bin/bash
# round 2, first GPU call: correctness of the LDS-staged search, A/B of its variants, the new bench line, a kernel-trace profile, the counter list
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c1
mkdir -p $O
for m in 1 2 0; do
  XEVE_HIP_ME_LDS=$m timeout 900 python -m pytest tests/test_hip_me.py tests/test_hip_inter.py tests/test_workload.py -x -q -m gpu > $O/pytest_lds$m.log 2>&1
  tail -3 $O/pytest_lds$m.log
done
for m in 0 1 2; do
  XEVE_HIP_ME_LDS=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_lds$m.json 2> $O/bench_lds$m.err
  cut -c1-600 $O/bench_lds$m.json
done
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err
cut -c1-3000 $O/bench_full.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_step -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1 )
tail -2 $O/prof_step.log
( cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1 )
wc -l $O/counters.txt
