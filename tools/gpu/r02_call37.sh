#!/bin/bash
mkdir -p gpurun_out
for mask in 0 1023 7 248 512 256; do
  echo "== zero mask $mask"
  XEVE_HIP_DEBUG_ZERO=$mask timeout 300 python -m pytest tests/test_hip_tree.py -x -q -k "host_form" 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -3
done > gpurun_out/r02_call37.log 2>&1
cat gpurun_out/r02_call37.log
