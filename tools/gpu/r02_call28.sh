#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/probe_tree.py --chains=1,512,2048 > gpurun_out/r02_tree_graph.log 2>&1
tail -5 gpurun_out/r02_tree_graph.log
timeout 600 python -m pytest tests/test_hip_tree.py -x -q 2>&1 | tail -3
