"""GPU suite: the drop-in dispatch tables of libxeve_hip.so (host pointers, one call per block, exactly the
reference's signatures) against the golden vectors and against the oracle on seeded inputs."""
import numpy as np
import pytest

from _golden_check import run_golden
from _libs import OracleTables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    from xeve_amd.tables import HipTables

    return HipTables(0)


@pytest.fixture(scope="module")
def O():
    return OracleTables()


def pels(r, shape, bd=10):
    return r.integers(0, 1 << bd, size=shape, dtype=np.int16)


def test_tables_match_golden_vectors(T):
    import xeve_amd

    before = xeve_amd.table_calls()
    assert run_golden(T) > 300
    assert xeve_amd.table_calls() - before > 300  # the answers came from the HIP table layer


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_distortions_every_table_entry(T, O, bd):
    r = np.random.default_rng(10 + bd)
    for lw in range(8):
        for lh in range(8):
            w, h = 1 << lw, 1 << lh
            s1, s2 = w + int(r.integers(0, 9)), w + int(r.integers(0, 9))
            a, b = pels(r, (h + 2, s1), bd), pels(r, (h + 2, s2), bd)
            if (lw + lh) % 3 == 0:
                a = (2 * a.astype(np.int32) - pels(r, a.shape, bd)).astype(np.int16)  # org_bi operand
            oa, ob = int(r.integers(0, s1 - w + 1)), int(r.integers(0, s2 - w + 1))
            args = (w, h, a, oa, b, ob, s1, s2, bd)
            assert T.sad(*args) == O.sad(*args), ("sad", w, h)
            assert T.ssd(*args) == O.ssd(*args), ("ssd", w, h)
            if w >= 2 and h >= 2:
                assert T.satd(*args) == O.satd(*args), ("satd", w, h)
            d0, d1 = np.full((h, w + 2), 3, np.int16), np.full((h, w + 2), 3, np.int16)
            T.diff(w, h, a, oa, b, ob, s1, s2, w + 2, d0, bd)
            O.diff(w, h, a, oa, b, ob, s1, s2, w + 2, d1, bd)
            assert np.array_equal(d0, d1), ("diff", w, h)


def test_distortion_edge_values(T, O):
    for w in (8, 64):
        a = np.full((w, w), 1023, np.int16)
        z = np.zeros((w, w), np.int16)
        for x, y in ((a, z), (z, a), (a, a), (z, z)):
            args = (w, w, x, 0, y, 0, w, w, 10)
            assert T.sad(*args) == O.sad(*args)
            assert T.ssd(*args) == O.ssd(*args)
            assert T.satd(*args) == O.satd(*args)
    # extreme of the declared domain: |a - b| = 32767
    a = np.full((8, 8), 16383, np.int16)
    b = np.full((8, 8), -16384, np.int16)
    assert T.sad(8, 8, a, 0, b, 0, 8, 8, 10) == O.sad(8, 8, a, 0, b, 0, 8, 8, 10)


@pytest.mark.parametrize("luma", [True, False])
def test_mc_every_phase_and_size(T, O, luma):
    from xeve_amd.device import baseline_coef_c, baseline_coef_l

    r = np.random.default_rng(20 + luma)
    coef = baseline_coef_l() if luma else baseline_coef_c()
    unit, nph = (16, 4) if luma else (32, 8)
    sizes = (4, 8, 16, 32, 64) if luma else (2, 4, 8, 16, 32)
    pad = 8
    for bd in (8, 10, 12):
        for w in sizes:
            for h in sizes:
                s_ref = w + 2 * pad + 1
                plane = pels(r, (h + 2 * pad, s_ref), bd)
                for px in range(nph):
                    for py in range(nph):
                        dx, dy = px * 4, py * 4
                        gx = (pad + int(r.integers(-2, 3))) * unit + dx
                        gy = (pad + int(r.integers(-2, 3))) * unit + dy
                        sp = w + 1
                        p0, p1 = np.full((h, sp), -7, np.int16), np.full((h, sp), -7, np.int16)
                        f = (T.mc_l, O.mc_l) if luma else (T.mc_c, O.mc_c)
                        f[0](dx, dy, plane, gx, gy, s_ref, sp, p0, w, h, bd, coef)
                        f[1](dx, dy, plane, gx, gy, s_ref, sp, p1, w, h, bd, coef)
                        assert np.array_equal(p0, p1), (luma, bd, w, h, dx, dy)


def test_mc_main_profile_style_full_coefficient_table(T, O):
    """The tables take the coefficient table as an argument (Main passes all 16 phases, xevem_mc.c:48);
    use a dense synthetic table to check that every row is honoured."""
    r = np.random.default_rng(33)
    coef = r.integers(-16, 64, size=(16, 8), dtype=np.int16)
    plane = pels(r, (40, 48))
    for dx in range(16):
        for dy in (0, 5, 11):
            gx, gy = 8 * 16 + dx, 8 * 16 + dy
            p0, p1 = np.zeros((16, 16), np.int16), np.zeros((16, 16), np.int16)
            T.mc_l(dx, dy, plane, gx, gy, 48, 16, p0, 16, 16, 10, coef)
            O.mc_l(dx, dy, plane, gx, gy, 48, 16, p1, 16, 16, 10, coef)
            assert np.array_equal(p0, p1), (dx, dy)


def test_average_and_recon(T, O):
    r = np.random.default_rng(40)
    for w in (4, 8, 16, 32, 64):
        a, b = pels(r, (w, w + 1)), pels(r, (w, w + 2))
        d0, d1 = np.zeros((w, w + 3), np.int16), np.zeros((w, w + 3), np.int16)
        T.avg(a, b, d0, w + 1, w + 2, w + 3, w, w)
        O.avg(a, b, d1, w + 1, w + 2, w + 3, w, w)
        assert np.array_equal(d0, d1)
        coef = r.integers(-2048, 2048, size=w * w, dtype=np.int16)
        pred = pels(r, w * w)
        for is_coef in (0, 1):
            r0, r1 = np.full((w, w + 4), 5, np.int16), np.full((w, w + 4), 5, np.int16)
            T.recon(coef, pred, is_coef, w, w, w + 4, r0, 10)
            O.recon(coef, pred, is_coef, w, w, w + 4, r1, 10)
            assert np.array_equal(r0, r1)


def test_transform_tables_1d_and_2d(T, O):
    r = np.random.default_rng(50)
    for log2n in range(1, 7):
        n = 1 << log2n
        for log2l in range(1, 7):
            line = 1 << log2l
            src = r.integers(-32767, 32768, size=n * line, dtype=np.int16)
            d0, d1 = np.zeros(n * line, np.int32), np.zeros(n * line, np.int32)
            T.tx(log2n, src, d0, 0, line, 0)
            O.tx(log2n, src, d1, 0, line, 0)
            assert np.array_equal(d0, d1), ("tx0", n, line)
            T.itx(log2n, src, d0, 0, line, 0)
            O.itx(log2n, src, d1, 0, line, 0)
            assert np.array_equal(d0, d1), ("itx0", n, line)
            s32 = r.integers(-(1 << 27), 1 << 27, size=n * line, dtype=np.int32)
            e0, e1 = np.zeros(n * line, np.int16), np.zeros(n * line, np.int16)
            sh = (log2l - 1 + 2) + (log2n + 6)
            T.tx(log2n, s32, e0, sh, line, 1)
            O.tx(log2n, s32, e1, sh, line, 1)
            assert np.array_equal(e0, e1), ("tx1", n, line)
            lim = (2**31 - 1) // (n * 90)
            s32 = r.integers(-lim, lim + 1, size=n * line, dtype=np.int32)
            T.itx(log2n, s32, e0, 17, line, 1)
            O.itx(log2n, s32, e1, 17, line, 1)
            assert np.array_equal(e0, e1), ("itx1", n, line)
    for bd in (8, 10):
        for lw in range(1, 7):
            for lh in range(1, 7):
                c0 = r.integers(-(1 << bd) + 1, 1 << bd, size=1 << (lw + lh), dtype=np.int16)
                c1 = c0.copy()
                T.trans(c0, lw, lh, bd)
                O.trans(c1, lw, lh, bd)
                assert np.array_equal(c0, c1), ("trans", lw, lh)
                T.itrans(c0, lw, lh, bd)
                O.itrans(c1, lw, lh, bd)
                assert np.array_equal(c0, c1), ("itrans", lw, lh)
