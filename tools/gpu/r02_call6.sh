#!/bin/bash
# round 2, call 6: candidate-per-lane motion search (correctness incl. the raster / integer-refinement branches inside the encoder, A/B)
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c6
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_me.py tests/test_hip_inter.py tests/test_workload.py tests/test_hip_rdo.py -x -q -m gpu > $O/pytest_cpl.log 2>&1
tail -12 $O/pytest_cpl.log
timeout 900 python -m pytest tests/test_integration_ref.py -x -q -m gpu -k "raster or whole_inter or motion_search" > $O/pytest_int.log 2>&1
tail -5 $O/pytest_int.log
for m in 0 1; do
  XEVE_HIP_ME_CPL=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_cpl$m.json 2> $O/bench_cpl$m.err
  python -c "
import json;d=json.load(open('$O/bench_cpl$m.json'));print($m, d['ms_per_step'], d['kernels_in_timed_region'], d['roofline']['algorithmic_GBps'])"
done
