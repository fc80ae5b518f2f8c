"""Pins the oracle: differential test of oracle/libxeve_oracle.so against the UNMODIFIED reference
compiled in place (oracle/_ref/libxeveb_ref.so), over the reference's plain-C, SSE4.1 and AVX2 dispatch
tables (reference: src_base/xeve_enc.c:722-779).  Skipped where oracle/_ref is absent; the committed
fixtures in tests/golden/ (made from the same reference build) cover that case.
"""
import ctypes as C

import numpy as np
import pytest

from _libs import ilog2, oracle, ptr, ref

pytestmark = pytest.mark.skipif(ref() is None, reason="oracle/_ref not built (needs /root/reference)")

PAD = 16  # border around blocks so that strided / filtered reads stay in-bounds


def rng(seed):
    return np.random.default_rng(seed)


def pels(r, shape, bd):
    return r.integers(0, 1 << bd, size=shape, dtype=np.int16)


SIZES = [1 << k for k in range(8)]


@pytest.mark.parametrize("variant", ["c", "sse", "avx"])
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_sad_ssd_diff_all_table_entries(variant, bd):
    O, V = oracle(), ref().variants[variant]
    r = rng(100 + bd)
    for w in SIZES:
        for h in SIZES:
            if variant != "c" and (w < 4 or h < 2 or w > 64 or h > 64):
                # SIMD tables carry NULL / partial entries outside the sizes the encoder issues
                continue
            s1, s2 = w + int(r.integers(0, 9)), w + int(r.integers(0, 9))
            a = pels(r, (h, s1), bd)
            b = pels(r, (h, s2), bd)
            if r.integers(0, 4) == 0:  # org_bi operand range: 2*org - pred  (xeve_pinter.c:143-156)
                a = (2 * a.astype(np.int32) - pels(r, (h, s1), bd)).astype(np.int16)
            lw, lh = ilog2(w), ilog2(h)
            f = V.sad[lw * 8 + lh]
            if not C.cast(f, C.c_void_p).value:
                continue
            assert f(w, h, ptr(a), ptr(b), s1, s2, bd) == O.xo_sad(w, h, ptr(a), ptr(b), s1, s2, bd), (w, h)
            f = V.ssd[lw * 8 + lh]
            if C.cast(f, C.c_void_p).value:
                assert f(w, h, ptr(a), ptr(b), s1, s2, bd) == O.xo_ssd(w, h, ptr(a), ptr(b), s1, s2, bd), (w, h)
            f = V.diff[lw * 8 + lh]
            if C.cast(f, C.c_void_p).value:
                sd = w + 3
                d0 = np.full((h, sd), 77, np.int16)
                d1 = d0.copy()
                f(w, h, ptr(a), ptr(b), s1, s2, sd, ptr(d0), bd)
                O.xo_diff(w, h, ptr(a), ptr(b), s1, s2, sd, ptr(d1))
                assert np.array_equal(d0, d1), (w, h)


def test_sad_full_int16_domain_matches_c_path():
    """The reference C path uses a 16-bit sign-mask abs on an int (xeve_util.h:55); the oracle restates it."""
    O, V = oracle(), ref().variants["c"]
    r = rng(7)
    a = r.integers(-32768, 32768, size=(64, 64), dtype=np.int16)
    b = r.integers(-32768, 32768, size=(64, 64), dtype=np.int16)
    assert V.sad[6 * 8 + 6](64, 64, ptr(a), ptr(b), 64, 64, 10) == O.xo_sad(64, 64, ptr(a), ptr(b), 64, 64, 10)


@pytest.mark.parametrize("variant", ["c", "sse"])
@pytest.mark.parametrize("bd", [8, 10])
def test_satd_all_shapes(variant, bd):
    O, V = oracle(), ref().variants[variant]
    r = rng(200 + bd)
    shapes = [(w, h) for w in (2, 4, 8, 16, 32, 64) for h in (2, 4, 8, 16, 32, 64)]
    for rep in range(6):
        for w, h in shapes:
            if variant == "sse" and (w < 4 or h < 4):
                continue
            so, sc = w + int(r.integers(0, 5)), w + int(r.integers(0, 5))
            o = pels(r, (h, so), bd)
            c = pels(r, (h, sc), bd)
            if rep == 0:
                c[:, :w] = o[:, :w]  # zero residual
            if rep == 1:
                o[:], c[:] = (1 << bd) - 1, 0  # max DC
            got = V.satd[0](w, h, ptr(o), ptr(c), so, sc, bd)
            exp = O.xo_satd(w, h, ptr(o), ptr(c), so, sc, bd)
            assert got == exp, (variant, w, h, rep, got, exp)


def _mc_case(r, w, h, bd, luma):
    taps = 8 if luma else 4
    s_ref = w + taps + 2 * PAD + int(r.integers(0, 4))
    rows = h + taps + 2 * PAD
    plane = pels(r, (rows, s_ref), bd)
    unit = 16 if luma else 32
    return plane, s_ref, PAD * unit, PAD * unit


@pytest.mark.parametrize("variant", ["c", "sse", "avx"])
@pytest.mark.parametrize("luma", [True, False])
def test_mc_all_phases(variant, luma):
    O, R = oracle(), ref()
    V = R.variants[variant]
    r = rng(300 + luma)
    tbl = V.mc_l if luma else V.mc_c
    coef = R.mc_l_coeff if luma else R.mc_c_coeff
    ocoef = O.mc_l_coeff if luma else O.mc_c_coeff
    assert bytes(coef) == bytes(ocoef)
    step, nph = (4, 4) if luma else (4, 8)
    sizes = (8, 16, 32, 64) if luma else (4, 8, 16, 32)
    for bd in (8, 10, 12):
        for w in sizes:
            for h in sizes:
                plane, s_ref, gx0, gy0 = _mc_case(r, w, h, bd, luma)
                for px in range(nph):
                    for py in range(nph):
                        dx, dy = px * step, py * step
                        gx = gx0 + int(r.integers(-3, 4)) * (16 if luma else 32) + dx
                        gy = gy0 + int(r.integers(-3, 4)) * (16 if luma else 32) + dy
                        sp = w + int(r.integers(0, 3))
                        p0 = np.full((h, sp), -5, np.int16)
                        p1 = p0.copy()
                        tbl[(dx != 0) * 2 + (dy != 0)](ptr(plane), gx, gy, s_ref, sp, ptr(p0), w, h, bd, coef)
                        (O.xo_mc_l if luma else O.xo_mc_c)(dx, dy, ptr(plane), gx, gy, s_ref, sp, ptr(p1), w, h, bd, ocoef)
                        assert np.array_equal(p0, p1), (variant, luma, bd, w, h, dx, dy)


@pytest.mark.parametrize("variant", ["c", "sse"])
def test_average(variant):
    O, V = oracle(), ref().variants[variant]
    r = rng(400)
    for w in (4, 8, 16, 32, 64):
        for h in (4, 8, 16, 32, 64):
            a, b = pels(r, (h, w), 10), pels(r, (h, w), 10)
            d0, d1 = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
            V.avg(ptr(a), ptr(b), ptr(d0), w, w, w, w, h)
            O.xo_avg(ptr(a), ptr(b), ptr(d1), w, w, w, w, h)
            assert np.array_equal(d0, d1)


def test_dct_matrices_match_reference_tables():
    O, R = oracle(), ref()
    for n in (2, 4, 8, 16, 32, 64):
        m = np.zeros((n, n), np.int8)
        O.xo_dct_matrix(n, ptr(m))
        assert np.array_equal(m, R.tm(n)), n


@pytest.mark.parametrize("variant", ["c"])
def test_1d_transforms(variant):
    """tx_pb*/itx_pb* with the shifts xeve_trans / xeve_itrans use (xeve_tq.c:396-404, xeve_itdq.c:435-440).
    Stress amplitudes: plain-C table only (the SIMD twins assume pipeline-range data and line >= 4; they are
    compared on real 2-D pipelines in test_2d_transforms)."""
    O, V = oracle(), ref().variants[variant]
    r = rng(500)
    bd = 10
    for log2n in range(1, 7):
        n = 1 << log2n
        for log2l in range(1, 7):
            line = 1 << log2l
            for amp in (1023, 32767):
                # forward, pass 1: s16 -> s32, shift 0
                src = r.integers(-amp, amp + 1, size=n * line, dtype=np.int16)
                d0, d1 = np.zeros(n * line, np.int32), np.zeros(n * line, np.int32)
                V.txb[log2n - 1](ptr(src), ptr(d0), 0, line, 0)
                O.xo_tx(log2n, ptr(src), ptr(d1), 0, line, 0)
                assert np.array_equal(d0, d1), ("tx0", variant, n, line)
                # forward, pass 2: s32 -> s16 with the 2-D shift
                sh = (log2l - 1 + bd - 8) + (log2n + 6)
                s32 = r.integers(-(amp << 12), (amp << 12) + 1, size=n * line, dtype=np.int32)
                e0, e1 = np.zeros(n * line, np.int16), np.zeros(n * line, np.int16)
                V.txb[log2n - 1](ptr(s32), ptr(e0), sh, line, 1)
                O.xo_tx(log2n, ptr(s32), ptr(e1), sh, line, 1)
                assert np.array_equal(e0, e1), ("tx1", variant, n, line)
                # inverse, pass 1 and pass 2
                V.itxb[log2n - 1](ptr(src), ptr(d0), 0, line, 0)
                O.xo_itx(log2n, ptr(src), ptr(d1), 0, line, 0)
                assert np.array_equal(d0, d1), ("itx0", variant, n, line)
                # The reference forms the inverse products/sums in 32-bit int (xeve_itdq.c:73-76,340-388:
                # `s8 * s32` terms summed as int, only then widened), so its defined domain for pass 2 is
                # |tb| <= (2^31-1)/(n*90); beyond that it is signed overflow (UB) and C/SSE/AVX2 disagree.
                lim = (2**31 - 1) // (n * 90)
                s32 = r.integers(-lim, lim + 1, size=n * line, dtype=np.int32)
                V.itxb[log2n - 1](ptr(s32), ptr(e0), 17, line, 1)
                O.xo_itx(log2n, ptr(s32), ptr(e1), 17, line, 1)
                assert np.array_equal(e0, e1), ("itx1", variant, n, line)


def _ref_trans(V, coef, lw, lh, bd):
    tb = np.zeros(64 * 64, np.int32)
    V.txb[lw - 1](ptr(coef), ptr(tb), 0, 1 << lh, 0)
    V.txb[lh - 1](ptr(tb), ptr(coef), (lw - 1 + bd - 8) + (lh + 6), 1 << lw, 1)


def _ref_itrans(V, coef, lw, lh, bd):
    tb = np.zeros(64 * 64, np.int32)
    V.itxb[lh - 1](ptr(coef), ptr(tb), 0, 1 << lw, 0)
    V.itxb[lw - 1](ptr(tb), ptr(coef), 7 + 12 - (bd - 8), 1 << lh, 1)


@pytest.mark.parametrize("variant", ["c", "sse", "avx"])
def test_2d_transforms(variant):
    O, V = oracle(), ref().variants[variant]
    r = rng(600)
    for bd in (8, 10):
        for lw in range(1, 7):
            for lh in range(1, 7):
                n = 1 << (lw + lh)
                resid = r.integers(-(1 << bd) + 1, 1 << bd, size=n, dtype=np.int16)
                a, b = resid.copy(), resid.copy()
                _ref_trans(V, a, lw, lh, bd)
                O.xo_trans(ptr(b), lw, lh, bd)
                assert np.array_equal(a, b), ("fwd", lw, lh)
                _ref_itrans(V, a, lw, lh, bd)
                O.xo_itrans(ptr(b), lw, lh, bd)
                assert np.array_equal(a, b), ("inv", lw, lh)


def test_recon():
    O, R = oracle(), ref()
    r = rng(700)
    for w in (4, 8, 16, 32, 64):
        coef = r.integers(-2048, 2048, size=w * w, dtype=np.int16)
        pred = pels(r, w * w, 10)
        for is_coef in (0, 1):
            s_rec = w + 5
            r0, r1 = np.full(w * s_rec, 9, np.int16), np.full(w * s_rec, 9, np.int16)
            R.recon(ptr(coef), ptr(pred), is_coef, w, w, s_rec, ptr(r0), 10)
            O.xo_recon(ptr(coef), ptr(pred), is_coef, w, w, s_rec, ptr(r1), 10)
            assert np.array_equal(r0, r1)
