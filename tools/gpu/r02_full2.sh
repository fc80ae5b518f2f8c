#!/bin/bash
# the whole GPU suite as the driver runs it + smoke(), with wall times
mkdir -p gpurun_out
( time timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r02_full2.log 2>&1
tail -30 gpurun_out/r02_full2.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02_smoke2.log 2>&1
tail -5 gpurun_out/r02_smoke2.log
