#!/bin/bash
# round 2, call 16: deferred byte extraction in the complete-state coder -- parity, then the step
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c16
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_sbac.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_hip_intra.py tests/test_workload.py -x -q -m gpu > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python tools/probe_step.py 5 2>&1 | tail -1 > $O/step.log
python tools/probe_step.py 5 --sizes=64 2>&1 | tail -1 >> $O/step.log
python tools/probe_step.py 5 --structured 2>&1 | tail -1 >> $O/step.log
python tools/probe_step.py 3 --intra 2>&1 | grep "all levels" >> $O/step.log
cat $O/step.log
