#!/usr/bin/env python3
"""junit XML of a `pytest tests -m gpu --junitxml=...` run on the GPU box -> tests/golden/gpu_suite_durations.json: seconds per test (set-up + call + tear-down), the
run's wall time, what failed or was skipped.  tests/test_gpu_suite_budget.py (CPU suite) holds the default GPU suite to 900 s with it.
usage: gpu_suite_durations.py run.xml [pytest.log] > tests/golden/gpu_suite_durations.json
       gpu_suite_durations.py --drop full.json "<node id>" ... > new.json   (cases since moved behind XEVE_GPU_FULL: removed with their seconds, and recorded)
       gpu_suite_durations.py --merge base.json part.xml > new.json   (tests added since the last full run, measured in a run of their own: their seconds are added
                                                                        to the full run's wall time and sum; tests the part repeats keep the full run's figure)"""
import json
import re
import sys
import xml.etree.ElementTree as ET

base = None
if sys.argv[1] == "--drop":  # gpu_suite_durations.py --drop full.json "<node id>" ... > new.json: cases of that run since moved behind XEVE_GPU_FULL leave with their seconds
    d = json.load(open(sys.argv[2]))
    gone = {k: d["tests"].pop(k) for k in sys.argv[3:]}
    d["wall_s"], d["sum_s"], d["passed"] = round(d["wall_s"] - sum(gone.values()), 1), round(d["sum_s"] - sum(gone.values()), 1), d["passed"] - len(gone)
    d["moved_behind_gpu_full_since"] = d.get("moved_behind_gpu_full_since", []) + [{"test": k, "seconds": v, "the_run_took_s": round(d["wall_s"] + sum(gone.values()), 1)} for k, v in gone.items()]
    d["skipped"] = d["skipped"] + list(gone)
    json.dump(d, sys.stdout, indent=1)
    print()
    sys.exit(0)
if sys.argv[1] == "--merge":
    base = json.load(open(sys.argv[2]))
    del sys.argv[1:3]
root = ET.parse(sys.argv[1]).getroot()
suite = root if root.tag == "testsuite" else root.find("testsuite")
tests, bad, skipped = {}, [], []
for tc in suite.iter("testcase"):
    cls, name = tc.get("classname", ""), tc.get("name", "")
    nodeid = cls.replace(".", "/") + ".py::" + name  # (module-level tests only: classname = the dotted module path)
    if tc.find("skipped") is not None:
        skipped.append(nodeid)
        continue
    tests[nodeid] = round(tests.get(nodeid, 0.0) + float(tc.get("time", 0.0)), 3)
    if tc.find("failure") is not None or tc.find("error") is not None:
        bad.append(nodeid)
wall = float(suite.get("time", 0.0))
if len(sys.argv) > 2:  # pytest's own last line ("356 passed, 298 deselected in 612.34s") is the figure that includes collection and imports
    m = re.findall(r" in ([0-9.]+)s", open(sys.argv[2]).read())
    if m:
        wall = max(wall, float(m[-1]))
if base is not None:
    new = {k: v for k, v in tests.items() if k not in base["tests"]}
    assert not bad, bad
    base["tests"].update(new)
    base["tests"] = dict(sorted(base["tests"].items()))
    base["wall_s"] = round(base["wall_s"] + sum(new.values()), 1)
    base["sum_s"] = round(base["sum_s"] + sum(new.values()), 1)
    base["passed"] += len(new)
    base["merged"] = base.get("merged", []) + [{"source": sys.argv[1], "tests": len(new), "seconds": round(sum(new.values()), 1)}]
    json.dump(base, sys.stdout, indent=1)
    print()
    sys.exit(0)
json.dump({"what": "seconds per test of the default GPU suite (`pytest tests -m gpu`, heavy repeats skipped: tests/conftest.py gpu_full) on one MI355X box",
           "source": sys.argv[1], "wall_s": round(wall, 1), "sum_s": round(sum(tests.values()), 1), "passed": len(tests) - len(bad), "failed": bad, "skipped": skipped,
           "tests": dict(sorted(tests.items()))}, sys.stdout, indent=1)
print()
