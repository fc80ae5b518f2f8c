#!/bin/bash
# round 2, call 5: unified bin-string counter + one-round RDO decision (correctness both forms, A/B), real-size encodes with the tables left to the reference
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c5
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_sbac.py tests/test_sbac_golden.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_workload.py tests/test_hip_mc_cu.py tests/test_hip_tables.py -x -q -m gpu > $O/pytest_spec.log 2>&1
tail -12 $O/pytest_spec.log
XEVE_HIP_RDO_SPEC=0 timeout 900 python -m pytest tests/test_hip_rdo.py tests/test_hip_inter.py tests/test_workload.py -x -q -m gpu > $O/pytest_nospec.log 2>&1
tail -5 $O/pytest_nospec.log
for m in 0 26000; do
  XEVE_HIP_RDO_SPEC=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_spec$m.json 2> $O/bench_spec$m.err
  python -c "
import json;d=json.load(open('$O/bench_spec$m.json'));print($m, d['ms_per_step'], d['kernels_in_timed_region'])"
done
timeout 1200 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -m gpu > $O/pytest_e2e.log 2>&1
tail -12 $O/pytest_e2e.log
