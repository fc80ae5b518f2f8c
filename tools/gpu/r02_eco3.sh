#!/bin/bash
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_hip_tree.py -x -q -k "b_picture_decided_and_written" 2>&1 | tail -12 > gpurun_out/r02_eco3.log
cat gpurun_out/r02_eco3.log
