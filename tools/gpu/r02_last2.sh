#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -3 > gpurun_out/r02_last2.log
cat gpurun_out/r02_last2.log
