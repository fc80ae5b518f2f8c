#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_integration_ref.py tests/test_hip_tree.py tests/test_hip_intra.py -x -q -k "ctu_mode_decision or hip_intra" 2>&1 | tail -25 > gpurun_out/r02_call21.log
cat gpurun_out/r02_call21.log
