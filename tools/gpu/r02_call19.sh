#!/bin/bash
# first run of the device-side CTU tree walk
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -25 > gpurun_out/r02_call19.log
cat gpurun_out/r02_call19.log
