#!/bin/bash
# round 2, call 10: the intra route inside the unmodified encoder
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c10
mkdir -p $O
timeout 1500 python -m pytest tests/test_integration_ref.py -x -q -m gpu -k "intra" --durations=8 > $O/pytest.log 2>&1
tail -25 $O/pytest.log
