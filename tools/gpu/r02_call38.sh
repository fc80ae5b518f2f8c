#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -3 > gpurun_out/r02_call38.log
cat gpurun_out/r02_call38.log
timeout 1500 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -k "cfg3_1920x1080_with_every_ctu" 2>&1 | tail -4 >> gpurun_out/r02_call38.log
tail -4 gpurun_out/r02_call38.log
