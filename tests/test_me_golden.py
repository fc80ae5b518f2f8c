"""CPU suite: the oracle's integer-pel diamond search against the committed results of the reference's own
me_ipel_diamond (tests/golden/me_v1.npz, made by tests/golden/make_me_golden.py)."""
from _me_cases import run_oracle
from _me_golden import golden_cases


def test_oracle_me_matches_reference_goldens():
    n = 0
    for c, (cost, mvx, mvy, beststep) in golden_cases():
        res = run_oracle(c)
        assert (res.cost, res.mv[0], res.mv[1], res.beststep) == (cost, mvx, mvy, beststep), (n, c["S"], c["bi"])
        n += 1
    assert n == 96


def test_oracle_spel_matches_reference_goldens():
    from _me_cases import run_oracle_spel
    from _me_golden import golden_spel_cases

    n = 0
    for c, (cost, mvx, mvy) in golden_spel_cases():
        res = run_oracle_spel(c)
        assert (res.cost, res.mv[0], res.mv[1]) == (cost, mvx, mvy), (n, c["S"], c["bi"])
        n += 1
    assert n == 64


def test_oracle_epzs_matches_reference_goldens():
    """pinter_me_epzs, every branch (plain, me_raster, me_ipel_refinement, bi), against tests/golden/me_epzs_v1.npz"""
    import numpy as np

    from _epzs_golden import GOLD, cases
    from _me_cases import run_oracle_epzs

    g = np.load(GOLD)["res"]
    cs = cases()
    assert len(cs) == len(g)
    kinds = set()
    for c, e in zip(cs, g):
        assert (c["S"], c["bi"], c["raster"], c["refi"], c["hpel_cnt"]) == tuple(int(v) for v in e[4:9])
        assert run_oracle_epzs(c, with_mot=True) == tuple(int(v) for v in e[:4]), (c["S"], c["bi"], c["raster"], c["hpel_cnt"])
        kinds.add((c["bi"], c["raster"], c["hpel_cnt"] == 0))
    assert len(kinds) >= 6
