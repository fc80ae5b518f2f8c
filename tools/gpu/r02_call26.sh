#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_e2e_real_sizes.py -x -q -s -k "every_ctu_decided" 2>&1 | tail -30 > gpurun_out/r02_call26.log
cat gpurun_out/r02_call26.log
