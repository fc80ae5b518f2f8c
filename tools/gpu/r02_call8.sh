#!/bin/bash
# round 2, call 8: Main-profile slice on the GPU (tables vs goldens / oracle, in-situ Main encode), the Baseline table tests after the staging refactor,
# and the speculative RDO decision for the 64x64 level only
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c8
mkdir -p $O
timeout 900 python -m pytest tests/test_main_profile.py tests/test_abi_symbols.py -x -q -m gpu --durations=8 > $O/pytest_main.log 2>&1
tail -15 $O/pytest_main.log
for spec in ; do
  XEVE_HIP_RDO_SPEC=$spec python tools/probe_step.py 5 2>&1 | tail -1
done > $O/spec.log 2>&1
cat $O/spec.log
