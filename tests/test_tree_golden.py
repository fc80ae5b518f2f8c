"""The oracle's CTU mode decision (xo_mode_analyze_ctu) against the committed golden fixture tests/golden/tree_v1.npz: CTUs of real encodes with what the REFERENCE made
of them (I, P and B pictures; made by tests/golden/make_tree_golden.py inside the unmodified encoder).  The pin that survives where /root/reference is absent."""
import numpy as np

from _tree_golden import load, run_oracle, same_as_reference


def test_oracle_ctu_mode_decision_matches_the_reference_goldens():
    kinds, modes = set(), set()
    for r in load():
        d, nb, m, mod = run_oracle(r)
        same_as_reference(r, d, nb, m, mod)
        kinds.add(r["slice_type"])
        modes |= set(np.unique(d["pred_mode"][0]).tolist())
    assert kinds == {0, 1, 2} or kinds == {0, 2} or kinds == {1, 2}, kinds
    assert len(modes) >= 3, modes


def test_golden_consumer_plumbing_with_the_oracle_as_engine():
    """the function the GPU test uses to feed a golden record to the library (run_walk: torch tensors, raw pointers, strides, the reference-picture table) run on CPU
    tensors with the oracle standing in for the library call"""
    import torch

    from _tree_golden import oracle_as_engine, run_walk

    for r in load():
        same_as_reference(r, *run_walk(r, torch.device("cpu"), oracle_as_engine(r)))
