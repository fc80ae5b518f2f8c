#!/bin/bash
mkdir -p gpurun_out
for seeds in 4302 4302,4301 4301,4301 4303,4304; do
  echo "== seeds $seeds"
  XEVE_TEST_HOST_FORM_SEEDS=$seeds timeout 300 python -m pytest tests/test_hip_tree.py -x -q -k "host_form" 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -3
done > gpurun_out/r02_call36.log 2>&1
cat gpurun_out/r02_call36.log
