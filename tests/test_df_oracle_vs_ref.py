"""Pins the oracle's deblocking / padding restatement against the UNMODIFIED reference driver (xeve_deblock -> xeve_deblock_tree ->
xeve_deblock_cu_ver / _cu_hor, xeve_picbuf_expand) through oracle/_ref/libref_df.so."""
import numpy as np
import pytest

from _df_cases import PAD, make_case, origin, tile_map
from _libs import oracle_df, ptr, ref_df

pytestmark = pytest.mark.skipif(ref_df() is None, reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("w,h,bd,idc,min_cu", [(128, 64, 10, 1, 4), (200, 136, 10, 1, 8), (64, 64, 8, 1, 4), (96, 72, 10, 0, 4), (72, 40, 12, 3, 4),
                                               (320, 192, 10, 1, 8), (8, 8, 10, 1, 4)])
def test_deblock_picture(w, h, bd, idc, min_cu):
    O, R = oracle_df(), ref_df()
    r = np.random.default_rng(w * 7 + h + bd + idc)
    for rep in range(3):
        c = make_case(r, w, h, bd, idc, min_cu)
        a = [p.copy() for p in c["planes"]]
        b = [p.copy() for p in c["planes"]]
        ms_a, ms_b, cm_b = c["map_scu"].copy(), c["map_scu"].copy(), c["map_cu_mode"].copy()  # (named: ptr() does not keep its array alive)
        O.xo_deblock_picture(ptr(a[0], origin(c, 0)), ptr(a[1], origin(c, 1)), ptr(a[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms_a),
                             ptr(c["map_cu_mode"]), ptr(c["refi"]), ptr(c["mv"]), c["p"])
        R.refdrv_deblock_picture(ptr(b[0], origin(c, 0)), ptr(b[1], origin(c, 1)), ptr(b[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms_b),
                                 ptr(cm_b), ptr(c["refi"]), ptr(c["mv"]), c["p"])
        for k in range(3):
            assert np.array_equal(a[k], b[k]), (rep, k)
        assert w * h < 4096 or any(not np.array_equal(a[k], c["planes"][k]) for k in range(3 if idc else 1))  # the filter did something


@pytest.mark.parametrize("w,h,sx,sy", [(256, 128, 2, 0), (256, 192, 0, 1), (320, 200, 3, 2), (200, 136, 1, 1)])
def test_deblock_picture_with_tiles(w, h, sx, sy):
    """two to four tiles: the reference's loop (xeve_deblock per tile and direction) leaves the edges between tiles alone -- the oracle with the tile map agrees,
    the map is the one the driver built, and the result differs from the one-tile filter (the tile edges are really skipped)"""
    O, R = oracle_df(), ref_df()
    r = np.random.default_rng(w + h + 17 * sx + 31 * sy)
    c = make_case(r, w, h, 10, 1, 8)
    a, b, one = ([p.copy() for p in c["planes"]] for _ in range(3))
    ms_a, ms_b, ms_c, cm_b = c["map_scu"].copy(), c["map_scu"].copy(), c["map_scu"].copy(), c["map_cu_mode"].copy()
    tid = tile_map(c, sx, sy)
    tid_ref = np.zeros_like(tid)
    R.refdrv_deblock_picture_tiles(ptr(b[0], origin(c, 0)), ptr(b[1], origin(c, 1)), ptr(b[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms_b), ptr(cm_b),
                                   ptr(c["refi"]), ptr(c["mv"]), c["p"], sx, sy, ptr(tid_ref))
    assert np.array_equal(tid, tid_ref) and tid.max() >= 1
    O.xo_deblock_picture_tiles(ptr(a[0], origin(c, 0)), ptr(a[1], origin(c, 1)), ptr(a[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms_a), ptr(c["map_cu_mode"]),
                               ptr(tid), ptr(c["refi"]), ptr(c["mv"]), c["p"])
    O.xo_deblock_picture(ptr(one[0], origin(c, 0)), ptr(one[1], origin(c, 1)), ptr(one[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms_c), ptr(c["map_cu_mode"]),
                         ptr(c["refi"]), ptr(c["mv"]), c["p"])
    for k in range(3):
        assert np.array_equal(a[k], b[k]), k
    assert any(not np.array_equal(a[k], one[k]) for k in range(3))


def test_picbuf_expand():
    O, R = oracle_df(), ref_df()
    r = np.random.default_rng(3)
    for (w, h, e) in [(64, 32, 16), (40, 24, 9), (8, 8, 16)]:
        s = w + 2 * PAD
        a = r.integers(0, 1024, size=(h + 2 * PAD, s)).astype(np.int16)
        b = a.copy()
        O.xo_picbuf_expand(ptr(a, PAD * s + PAD), s, w, h, e)
        z = np.zeros(4, np.int16)
        R.refdrv_picbuf_expand(ptr(b, PAD * s + PAD), ptr(z), ptr(z), s, 0, w, h, 0, 0, e, 0, 0)
        assert np.array_equal(a, b)
