"""CPU suite: the oracle's Main-profile restatements (xo_mc_main, xo_tx / xo_itx step 2) against the committed outputs of the reference's own
Main-profile tables (tests/golden/main_v1.npz, made by tests/golden/make_main_golden.py) -- the pin that travels where oracle/_ref cannot."""
from _main_cases import OracleMain, check_golden


def test_oracle_main_slice_matches_golden():
    assert check_golden(OracleMain()) > 200000
