#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -12 > gpurun_out/r02_call31.log
cat gpurun_out/r02_call31.log
timeout 900 python tools/probe_tree.py --chains=1,512,2048,8192 > gpurun_out/r02_tree_lane.log 2>&1
tail -5 gpurun_out/r02_tree_lane.log
