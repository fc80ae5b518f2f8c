#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_tree.py -x -q -k "host_form" 2>&1 | tail -12 > gpurun_out/r02_call35.log
cat gpurun_out/r02_call35.log
timeout 1500 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -k "cfg3_1920x1080_with_every_ctu" 2>&1 | tail -6 >> gpurun_out/r02_call35.log
tail -6 gpurun_out/r02_call35.log
