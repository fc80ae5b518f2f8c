"""Pins the oracle's integer-pel diamond search (xo_me_ipel_diamond, xo_mv_bits) against the REAL static functions of
the reference (src_base/xeve_pinter.c:74-120, 363-551), reached through oracle/ref_me_driver.c, which compiles the
reference's xeve_pinter.c in place."""
import ctypes as C

import numpy as np
import pytest

from _libs import oracle_me, ptr, ref_me
from ctypes import c_int, c_void_p
from _me_cases import PAD, REFI_BITS_2_0, make_case, run_oracle

pytestmark = pytest.mark.skipif(ref_me() is None, reason="oracle/_ref/libref_me.so not built (needs /root/reference)")


def test_mv_bits_table_closed_form():
    O, R = oracle_me(), ref_me()
    assert R.refdrv_refi_bits(2, 0) == REFI_BITS_2_0
    for mvd in list(range(-2100, 2101)) + [-5000, 5000, -32768, 32767, 4095, 4096, -4096, 8191, 8192, 16383, 16384]:
        assert O.xo_mv_bits(mvd, 0) + R.refdrv_refi_bits(2, 1) == R.refdrv_mv_bits(mvd, 0, 2, 1), mvd
        assert O.xo_mv_bits(3, mvd) + R.refdrv_refi_bits(1, 0) == R.refdrv_mv_bits(3, mvd, 1, 0), mvd


def run_ref(c):
    R = ref_me()
    lg = c["S"].bit_length() - 1
    out = (C.c_int * 4)()
    org0, ref0 = ptr(c["org"], PAD * c["s"] + PAD), ptr(c["ref"], PAD * c["s"] + PAD)
    rng = np.array(c["range"], np.int16)
    gmvp, mvi = np.array(c["gmvp"], np.int16), np.array(c["mvi"], np.int16)
    mn, mx = np.array(c["min_clip"], np.int32), np.array(c["max_clip"], np.int32)
    # gop_size / poc chosen so that get_range_ipel derives exactly c["sr"]: (msr * |poc - ref_poc| + gop/2) / gop = sr
    gop, poc, ref_poc = c["msr"], c["sr"], 0
    cost = R.refdrv_me_ipel_diamond(org0, c["s"], ptr(c["org_bi"]), ref0, c["s"], c["x"], c["y"], lg, lg, 10, ptr(rng), ptr(gmvp), ptr(mvi),
                                    c["bi"], c["faststep"], c["lambda_mv"], 2, 0, c["mot_other"], c["msr"], gop, poc, ref_poc, ptr(mn), ptr(mx),
                                    c["beststep_in"], out)
    return cost, out[0], out[1], out[2], out[3]


@pytest.mark.parametrize("bi", [0, 1, 2])
@pytest.mark.parametrize("textured", [False, True])
def test_me_ipel_diamond_matches_reference(bi, textured):
    r = np.random.default_rng(900 + bi * 2 + textured)
    steps = set()
    for it in range(60):
        c = make_case(r, int(r.choice([8, 16, 32, 64])), bi, textured)
        cost, mvx, mvy, beststep, mot = run_ref(c)
        res = run_oracle(c)
        assert (res.cost, res.mv[0], res.mv[1], res.beststep) == (cost, mvx, mvy, beststep), (it, c["S"], c["x"], c["y"])
        if bi != 1 and res.best_mv_bits > 0:
            assert mot == res.best_mv_bits
        steps.add(beststep)
    if textured and bi != 1:
        assert max(steps) >= 4  # the diamond rings were actually exercised


def run_ref_spel(c):
    from _libs import ref_spel

    R = ref_spel()
    lg = c["S"].bit_length() - 1
    out = (C.c_int * 3)()
    gmvp, mvi = np.array(c["gmvp"], np.int16), np.array(c["mvi"], np.int16)
    cost = R.refdrv_me_spel_pattern(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], c["x"], c["y"],
                                    lg, lg, 10, ptr(gmvp), ptr(mvi), c["bi"], c["lambda_mv"], 2, 0, c["mot_other"], c["hpel_cnt"], c["qpel_cnt"], out)
    return cost, out[0], out[1], out[2]


@pytest.mark.parametrize("bi", [0, 1])
@pytest.mark.parametrize("textured", [False, True])
def test_me_spel_pattern_matches_reference(bi, textured):
    from _me_cases import make_planes, make_spel_job, run_oracle_spel

    r = np.random.default_rng(950 + bi * 2 + textured)
    pl = make_planes(r, textured)
    for it in range(60):
        c = make_spel_job(r, pl, int(r.choice([8, 16, 32, 64])), bi)
        cost, mvx, mvy, mot = run_ref_spel(c)
        res = run_oracle_spel(c)
        assert (res.cost, res.mv[0], res.mv[1]) == (cost, mvx, mvy), (it, c["S"], c["hpel_cnt"], c["qpel_cnt"])
        if not bi and res.best_mv_bits > 0:
            assert mot == res.best_mv_bits


@pytest.mark.parametrize("bi", [0, 1])
@pytest.mark.parametrize("textured", [False, True])
def test_me_epzs_matches_reference(bi, textured):
    """the whole per-list search of pinter_me_epzs (me_complexity 1): diamond, refinement loop, sub-pel pattern"""
    from _libs import ref_epzs
    from _me_cases import make_epzs_job, make_planes, run_oracle_epzs

    R = ref_epzs()
    r = np.random.default_rng(970 + bi * 2 + textured)
    pl = make_planes(r, textured)
    moved = 0
    for it in range(50):
        c = make_epzs_job(r, pl, int(r.choice([8, 16, 32, 64])), bi)
        lg = c["S"].bit_length() - 1
        mvp, mv = np.array(c["mvp"], np.int16), np.array(c["mv0"], np.int16)
        mn, mx = np.array(c["min_clip"], np.int32), np.array(c["max_clip"], np.int32)
        cost = R.refdrv_me_epzs(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], c["x"], c["y"], lg, lg,
                                10, ptr(mvp), ptr(mv), bi, c["lambda_mv"], 2, 0, c["mot_other"], c["msr"], c["msr"], c["sr"], 0, ptr(mn), ptr(mx),
                                c["hpel_cnt"], c["qpel_cnt"])
        R.refdrv_me_epzs_mot_bits.restype = C.c_int
        assert run_oracle_epzs(c, with_mot=True) == (cost, int(mv[0]), int(mv[1]), R.refdrv_me_epzs_mot_bits()), (it, c["S"])
        moved += (int(mv[0]), int(mv[1])) != tuple(c["mvp"])
    assert moved > 25


def test_epzs_with_raster_search_and_integer_refinement_matches_reference():
    """the branches of pinter_me_epzs the presets fast / medium do not take: me_raster (me_complexity > 1: placebo) after a first search that ended far
    from its start, with the step scaled by refi + 1, and me_ipel_refinement in place of the sub-pel pattern (me_level = ME_LEV_IPEL)"""
    from _me_cases import make_epzs_job, make_planes, run_oracle_epzs

    R = ref_me()
    R.refdrv_me_epzs_x.restype = C.c_uint32
    R.refdrv_me_epzs_x.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, C.c_uint32] + [c_int] * 7 + \
                                  [c_void_p, c_void_p, c_int, c_int, c_int]
    R.refdrv_me_epzs_mot_bits.restype = C.c_int
    r = np.random.default_rng(2024)
    rastered = refined = 0
    for it in range(160):
        pl = make_planes(r, textured=it % 4 != 0)
        S, bi = int(r.choice([8, 16, 32, 64])), int(r.choice([0, 0, 0, 1]))
        c = make_epzs_job(r, pl, S, bi)
        c["raster"], c["refi"] = int(r.random() < 0.75), int(r.integers(0, 2))
        if r.random() < 0.4:
            c["hpel_cnt"], c["qpel_cnt"] = 0, 0
            refined += 1
        c["mvp"] = (int(r.integers(-160, 161)), int(r.integers(-160, 161)))  # far from the true motion: the first search walks, beststep grows
        lg = S.bit_length() - 1
        mvp, mv = np.array(c["mvp"], np.int16), np.array(c["mv0"], np.int16)
        mn, mx = np.array(c["min_clip"], np.int32), np.array(c["max_clip"], np.int32)
        cost = R.refdrv_me_epzs_x(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], c["x"], c["y"], lg, lg, 10,
                                  ptr(mvp), ptr(mv), bi, c["lambda_mv"], 2, c["refi"], c["mot_other"], c["msr"], c["msr"], c["sr"], 0, ptr(mn), ptr(mx), c["hpel_cnt"],
                                  c["qpel_cnt"], 2 if c["raster"] else 1)
        got = run_oracle_epzs(c, with_mot=True)
        assert got == (cost, int(mv[0]), int(mv[1]), R.refdrv_me_epzs_mot_bits()), (it, S, bi, c["raster"], c["refi"], c["hpel_cnt"], got, cost, mv)
        c0 = dict(c, raster=0)
        rastered += c["raster"] and bi == 0 and run_oracle_epzs(c0, with_mot=True) != got
    assert rastered > 10 and refined > 30, (rastered, refined)
