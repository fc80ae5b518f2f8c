#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -30 > gpurun_out/r02_call24.log
cat gpurun_out/r02_call24.log
