#!/bin/bash
mkdir -p gpurun_out
{ timeout 600 python tools/probe_tree.py --chains=1,64,512,2048; timeout 300 python tools/probe_tree.py --chains=1,512 --content=smooth; } > gpurun_out/r02_tree_chains.log 2>&1
cat gpurun_out/r02_tree_chains.log
