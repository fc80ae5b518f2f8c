#!/bin/bash
# round 2, call 2: register-context CABAC automaton (correctness + A/B), resident-picture e2e at small size and 1280x720 / 1920x1080
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c2
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_sbac.py tests/test_sbac_golden.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_workload.py -x -q -m gpu > $O/pytest_reg.log 2>&1
tail -3 $O/pytest_reg.log
for m in 0 1; do
  XEVE_HIP_SBAC_REG=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_reg$m.json 2> $O/bench_reg$m.err
  python -c "
import json;d=json.load(open('$O/bench_reg$m.json'));print($m, d['ms_per_step'], d['kernels_in_timed_region'])"
done
timeout 900 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -m gpu -k "small or 720" > $O/pytest_e2e_a.log 2>&1
tail -8 $O/pytest_e2e_a.log
timeout 1200 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -m gpu -k "1080" > $O/pytest_e2e_b.log 2>&1
tail -5 $O/pytest_e2e_b.log
