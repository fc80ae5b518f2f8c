"""Pins xo_inter_candidates (the merge / MVP candidates and the collocated vector an inter job carries) against the reference's exported
xeve_get_avail_inter + xeve_get_motion (xeve_util.c:652-714, 526-573) on random per-unit maps: coded / intra / IBC flags, two tiles, picture borders."""
import numpy as np
import pytest

from _inter_cases import make_maps
from _libs import INTER_JOB_DTYPE, oracle_cand, ptr, ref_cand

pytestmark = pytest.mark.skipif(ref_cand() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("slice_type,tiles", [(0, 1), (1, 1), (0, 2)])
def test_inter_candidates(slice_type, tiles):
    O, R = oracle_cand(), ref_cand()
    r = np.random.default_rng(31 + slice_type + tiles)
    w_scu, h_scu = 48, 32
    map_scu, tidx, map_mv, c0, c1 = make_maps(r, w_scu, h_scu, tiles)
    unavailable = 0
    for lw in (3, 4, 5, 6):
        s = 1 << (lw - 2)
        for _ in range(300):
            a, b = np.zeros(1, INTER_JOB_DTYPE), np.zeros(1, INTER_JOB_DTYPE)
            x = int(r.integers(0, w_scu // s)) * s * 4
            y = int(r.integers(0, h_scu // s)) * s * 4
            a["x"], a["y"], b["x"], b["y"] = x, y, x, y
            O.xo_inter_candidates(ptr(map_scu), ptr(tidx), ptr(map_mv), ptr(c0), ptr(c1), w_scu, h_scu, lw, lw, slice_type, ptr(a))
            R.refdrv_inter_candidates(ptr(map_scu), ptr(tidx), ptr(map_mv), ptr(c0), ptr(c1), w_scu, h_scu, lw, lw, slice_type, ptr(b))
            assert a.tobytes() == b.tobytes(), (lw, x, y, a, b)
            unavailable += int((a["mvp"][0, 0, :3] == 1).all(axis=1).sum())
    assert unavailable > 300
