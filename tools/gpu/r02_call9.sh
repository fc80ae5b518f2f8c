#!/bin/bash
# round 2, call 9: the 72-model coder state everywhere + the intra analysis on the GPU
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c9
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_intra.py tests/test_hip_sbac.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_rdoq.py tests/test_abi_symbols.py -x -q -m gpu --durations=8 > $O/pytest.log 2>&1
tail -25 $O/pytest.log
