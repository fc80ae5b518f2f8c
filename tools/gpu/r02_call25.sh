#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_integration_ref.py -x -q -s -k "every_ctu_decided or ctu_mode_decision_on_the_gpu" 2>&1 | tail -30 > gpurun_out/r02_call25.log
cat gpurun_out/r02_call25.log
