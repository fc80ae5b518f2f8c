#!/bin/bash
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_hip_tree.py -q 2>&1 | tail -4 > gpurun_out/r02_tree_all.log
cat gpurun_out/r02_tree_all.log
