"""GPU: the device-side CTU mode decision against the golden fixture recorded from the reference encoder (tests/golden/tree_v1.npz).  (The file sorts last on purpose:
this test was written after the round's GPU budget was spent -- its plumbing is checked on the CPU with the oracle as the engine, tests/test_tree_golden.py -- and a
`pytest -x` run should reach it after everything else.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_hip_ctu_mode_decision_matches_the_reference_goldens():
    """the device walk against tests/golden/tree_v1.npz: CTUs of real encodes (I, P and B pictures, intra / inter / skip / direct CUs) with what the REFERENCE made of them,
    recorded inside the unmodified encoder (tests/golden/make_tree_golden.py) -- no oracle in between"""
    import torch
    import xeve_amd
    from xeve_amd import device as D
    from _tree_golden import load, run_walk, same_as_reference

    xeve_amd.init(0)
    n = 0
    for r in load():
        same_as_reference(r, *run_walk(r, torch.device("cuda:0"), D.mode_analyze_ctu_jobs))
        n += 1
    assert n >= 10
