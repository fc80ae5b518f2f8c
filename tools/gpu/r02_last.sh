#!/bin/bash
# last sanity of the round: the tree tests, the every-CTU integration tests, the bench line
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_tree.py tests/test_integration_ref.py -x -q -k "hip_ctu or every_ctu_decided" 2>&1 | tail -3 > gpurun_out/r02_last.log
timeout 600 python bench.py > gpurun_out/r02_last_bench.json 2>> gpurun_out/r02_last.log
python - <<'PY' >> gpurun_out/r02_last.log
import json
d=json.loads(open('gpurun_out/r02_last_bench.json').read().strip().split("\n")[-1])
print("value",d["value"],d["unit"],"ms_per_step",d["ms_per_step"],"roofline frac",d["roofline"]["frac"])
print("ctu walk",d["secondary"]["ctu_mode_decision_I_pictures"])
print("cpu",d["cpu_baseline"]["value"])
PY
cat gpurun_out/r02_last.log
