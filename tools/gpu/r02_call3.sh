#!/bin/bash
# round 2, call 3: bin-stream CABAC counter (correctness + A/B), graph-replayed per-CU calls in the encoder (720p timing)
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_sbac.py tests/test_sbac_golden.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_workload.py tests/test_hip_df.py -x -q -m gpu > $O/pytest_a.log 2>&1
tail -15 $O/pytest_a.log
for m in 0 1; do
  XEVE_HIP_SBAC_STREAM=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_s$m.json 2> $O/bench_s$m.err
  python -c "
import json;d=json.load(open('$O/bench_s$m.json'));print($m, d['ms_per_step'], d['kernels_in_timed_region'])"
done
timeout 900 python -m pytest tests/test_e2e_real_sizes.py tests/test_integration_ref.py -x -q -s -m gpu -k "small or 720 or shard or whole_inter" > $O/pytest_e2e.log 2>&1
tail -12 $O/pytest_e2e.log
