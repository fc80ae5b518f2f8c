#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -12 > gpurun_out/r02_call32.log
cat gpurun_out/r02_call32.log
timeout 300 python tools/probe_tree.py --chains=1,2048 2>&1 | tail -2
