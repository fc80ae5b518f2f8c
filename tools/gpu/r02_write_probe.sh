#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/probe_tree.py --chains=1,2048 --write 2>&1 | grep -E "chains|Error|error" > gpurun_out/r02_tree_write.log
cat gpurun_out/r02_tree_write.log
