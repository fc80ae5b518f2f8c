#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -12 > gpurun_out/r02_call29.log
cat gpurun_out/r02_call29.log
