"""CPU suite: the default GPU suite has to fit the driver's window (1200 s on one MI355X; round 4's was killed there after 31 of 314 tests).  The per-test seconds of the
last full run on the GPU box are committed (tests/golden/gpu_suite_durations.json, tools/gpu_suite_durations.py); this test fails when they add up to more than 900 s,
when that run had a failure, or when the suite has since gained a GPU test the file has never seen (run the suite on the GPU and regenerate the file)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUR = os.path.join(ROOT, "tests", "golden", "gpu_suite_durations.json")
BUDGET_S = 900.0

COLLECT = r"""
import json, os, sys
import pytest
class P:
    ids = []
    def pytest_collection_modifyitems(self, session, config, items):
        P.ids = [(i.nodeid, i.get_closest_marker("gpu_full") is not None) for i in items if i.get_closest_marker("gpu") is not None]
pytest.main(["--collect-only", "-q", "-p", "no:cacheprovider", os.path.join(sys.argv[1], "tests")], plugins=[P()])
sys.stdout.write("\nIDS=" + json.dumps(P.ids) + "\n")
"""


def _collected():
    env = {k: v for k, v in os.environ.items() if k != "XEVE_GPU_FULL"}
    p = subprocess.run([sys.executable, "-c", COLLECT, ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("IDS=")]
    assert line, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(line[-1][4:])


def test_the_default_gpu_suite_fits_the_drivers_window():
    d = json.load(open(DUR))
    assert not d["failed"], d["failed"]
    assert d["wall_s"] <= BUDGET_S and d["sum_s"] <= BUDGET_S, (d["wall_s"], d["sum_s"])
    ids = _collected()
    default = [n for n, full in ids if not full]
    heavy = [n for n, full in ids if full]
    assert len(default) >= 300 and heavy  # (the leaf-kernel parity tests alone are ~250)
    unseen = [n for n in default if n not in d["tests"] and n not in d["skipped"]]  # (a test that skips itself at run time costs nothing)
    assert not unseen, "GPU tests without a measured duration (run the suite on the GPU box, then tools/gpu_suite_durations.py): %s" % unseen[:8]
    assert not [n for n in heavy if n in d["tests"]], "a gpu_full case ran in the default suite"


def test_the_row_defining_tests_run_before_the_real_size_encodes():
    """collection order = run order: every leaf-kernel file before the in-encoder routes, the cases at BASELINE's picture sizes last"""
    ids = [n for n, _ in _collected()]
    pos = {}
    for i, n in enumerate(ids):
        pos.setdefault(n.split("::")[0].split("/")[-1], []).append(i)
    first, last = (lambda f: pos[f][0]), (lambda f: pos[f][-1])
    for leaf in ("test_hip_tables.py", "test_hip_batched.py", "test_rdoq.py", "test_hip_mc_cu.py", "test_hip_me.py", "test_hip_sbac.py", "test_hip_rdo.py", "test_hip_df.py"):
        assert last(leaf) < first("test_integration_ref.py") < first("test_e2e_real_sizes.py"), leaf
    real = [i for i, n in enumerate(ids) if "at_real_picture_sizes" in n or "full_eight_frame" in n]
    assert real and min(real) > last("test_e2e_real_sizes.py") and max(real) == len(ids) - 1
