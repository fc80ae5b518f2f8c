#!/bin/bash
mkdir -p gpurun_out
{ timeout 300 python tools/probe_step.py 8 --graph; timeout 300 python tools/probe_step.py 8 --graph --1080p; timeout 300 python tools/probe_step.py 8 --graph --structured; } > gpurun_out/r02_eager_vs_graph.log 2>&1
tail -12 gpurun_out/r02_eager_vs_graph.log
