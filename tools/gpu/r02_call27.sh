#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/probe_tree.py --chains=1,512,2048,8192 > gpurun_out/r02_tree_chains2.log 2>&1
tail -5 gpurun_out/r02_tree_chains2.log
timeout 900 python bench.py > gpurun_out/r02_bench_tree.json 2> gpurun_out/r02_bench_tree.err
tail -c 3000 gpurun_out/r02_bench_tree.json; tail -5 gpurun_out/r02_bench_tree.err
