"""Both CTU walks stay pinned on the GPU whatever the library's own choice is (walk.hip: the fused kernel up to 1024 chains in lockstep, the composed walk above -- the
suite's own batches are all narrow): the same reference-bitstream and CTU-tree tests once more in a process with the composed walk pinned (XEVE_HIP_WALK=0) and in one
with the fused walk carrying three chains per team (XEVE_HIP_WALK=1 XEVE_HIP_WALK_C=3: teams of several chains are otherwise only formed beyond 1024 chains)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PICK = "tiny_ldb_fast or tiny_ra_medium or moving_ldb_ref3 or jumpy_ldb_fast or gops_128x64_noise or gops_cif_noise_m8 or the_same_as_its_gops"


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"XEVE_HIP_WALK": "0"}, {"XEVE_HIP_WALK": "1", "XEVE_HIP_WALK_C": "3"}], ids=["composed", "fused_3_chains_per_team"])
def test_the_encoder_and_tree_tests_with_the_walk_pinned(env):
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_enc_gpu.py"), os.path.join(ROOT, "tests", "test_hip_tree.py"),
           "-k", PICK + " or tree"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, **env))
    tail = p.stdout[-1500:]
    assert p.returncode == 0, (tail, p.stderr[-1500:])
    assert " passed" in tail and "failed" not in tail, tail
