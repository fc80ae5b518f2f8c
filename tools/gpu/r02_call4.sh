#!/bin/bash
# round 2, call 4: profile with the bin-stream counter (BINK 32), where the time goes per level, e2e per-CU split
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c4
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['kernels_in_timed_region'], d['kernels'], d['secondary'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 > $O/prof_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_struct -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 3 --structured > $O/prof_struct.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -m gpu -k "small or 720 or 3840" > $O/pytest_e2e.log 2>&1
tail -8 $O/pytest_e2e.log
