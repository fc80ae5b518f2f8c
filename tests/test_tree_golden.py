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


def test_writers_match_the_reference_goldens():
    """the reference WRITER's side of the recorded CTUs (xeve_eco_tree: the coder state it left, the bytes it put into the bitstream, the unit flags) against the oracle's
    xo_eco_ctu and against the host side of the lane writer the GPU runs (xeve_amd/csrc/eco_lane.h)"""
    import ctypes as C

    import _lane
    from _libs import c_int, c_void_p, oracle, ptr
    from _tree_golden import writer_inputs, writer_same_as_reference

    O = oracle()
    O.xo_eco_ctu.restype = c_int
    O.xo_eco_ctu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_void_p, c_int]
    n, kinds, total = 0, set(), 0
    for r in load():
        if "wr" not in r:
            continue
        nr = np.array(r["wr"]["num_refp"], np.int32)
        d, st, m = writer_inputs(r)
        by = np.zeros(1 << 16, np.uint8)
        k = O.xo_eco_ctu(ptr(st), ptr(d), C.addressof(r["P"]), ptr(nr), ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]), ptr(m["cu_mode"]), r["x0"], r["y0"], ptr(by), by.size)
        writer_same_as_reference(r, st, by[:k], m)
        if _lane.available():
            L, sc = _lane.lane(), _lane.scans()
            d, st, m = writer_inputs(r)
            m["scu"] |= np.uint32(1 << 31) * 0  # (the lane writer resets the CTU's coded flags itself; they already are)
            by2 = np.zeros(1 << 16, np.uint8)
            k2 = L.xl_host_eco_ctu(r["idc"], r["slice_type"], r["P"].log2_ctu, r["w"], r["h"], r["w"] // 4, int(nr[0]), int(nr[1]), ptr(sc[0]), ptr(sc[1]), ptr(sc[2]), ptr(st), ptr(d),
                                   ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]), ptr(m["cu_mode"]), r["x0"], r["y0"], ptr(by2), by2.size)
            writer_same_as_reference(r, st, by2[:k2], m)
        n, total = n + 1, total + k
        kinds.add(r["slice_type"])
    assert n >= 8 and len(kinds) >= 2 and total > 100, (n, kinds, total)
