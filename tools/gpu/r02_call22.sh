#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -k "i_picture_decided" 2>&1 | tail -25 > gpurun_out/r02_call22.log
cat gpurun_out/r02_call22.log
