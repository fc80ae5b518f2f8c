#!/bin/bash
# round 2, call 11: intra phase of the workload (test + timing + kernel profile), counts of the in-situ intra route
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c11
mkdir -p $O
timeout 600 python -m pytest tests/test_workload.py -x -q -m gpu -k intra > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python tools/probe_step.py 3 --intra > $O/intra_iid.log 2>&1; cat $O/intra_iid.log
python tools/probe_step.py 3 --intra --structured > $O/intra_struct.log 2>&1; cat $O/intra_struct.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_intra -o st -- python $GRAFT_REPO_ROOT/tools/probe_step.py 1 --intra > $GRAFT_REPO_ROOT/$O/prof_intra.log 2>&1 )
python - <<'PY'
import sys, re
sys.path.insert(0, "tests")
from _e2e import CASES, make_yuv, run_app
for name in ("moving_cif_allintra_fast", "noise_allintra_medium", "moving_ra_medium"):
    w, h, n, seed, extra = CASES[name]
    make_yuv("/tmp/a.yuv", w, h, n, seed)
    import time
    t = time.time()
    md5, size, err = run_app("/tmp/a.yuv", "/tmp/a.evc", w, h, n, extra, hip=True, timeout=3000, intra=True, tables=False)
    print(name, round(time.time() - t, 2), "s", [l for l in err.splitlines() if "intra" in l])
PY
