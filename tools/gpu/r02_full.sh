#!/bin/bash
# the whole GPU suite as the driver runs it, with durations; then smoke()
set -x
export TMPDIR=/tmp
O=gpurun_out/r02full
mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
