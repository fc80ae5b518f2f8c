"""GPU suite: the batched device API (planes resident in HBM, many blocks per launch) against the oracle on
seeded inputs, and -- at the full 3840x2160 size of the headline config -- through size-independent properties."""
import numpy as np
import pytest

from _libs import oracle, ptr

pytestmark = pytest.mark.gpu

PAD = 144  # luma padding of a reference picture plane (reference: src_base/xeve_def.h:380)


@pytest.fixture(scope="module")
def dev():
    import torch

    import xeve_amd

    xeve_amd.init(0)
    return torch.device("cuda:0")


def make_planes(r, W, H, bd=10, pad=PAD):
    s = W + 2 * pad
    org = r.integers(0, 1 << bd, size=(H + 2 * pad, s), dtype=np.int16)
    ref = r.integers(0, 1 << bd, size=(H + 2 * pad, s), dtype=np.int16)
    return org, ref, s


def diamond(s_ref, rng_steps=(4, 8, 16, 32, 64)):
    """candidate geometry of one me_ipel_diamond pass (reference: src_base/xeve_pinter.c:363-551): dense 5x5,
    then 4-point / 8-point / 16-point diamonds at growing L1 radius"""
    c = [(dx, dy) for dy in range(-2, 3) for dx in range(-2, 3)]
    for st in rng_steps:
        if st == 4:
            c += [(0, -4), (-4, 0), (4, 0), (0, 4), (0, 0)]
        else:
            n = 8 if st == 8 else 16
            q = n // 4
            for i in range(n):
                a, b = i % q, i // q
                dx, dy = [(a, -(q - a)), (q - a, a), (-a, q - a), (-(q - a), -a)][b]
                c.append((dx * st // q, dy * st // q))
            c.append((0, 0))
    return np.array([dy * s_ref + dx for dx, dy in c], np.int32), c


@pytest.mark.parametrize("size", [8, 16, 32, 64])
@pytest.mark.parametrize("signed", [False, True])
def test_sad_jobs_vs_oracle(dev, size, signed):
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(100 + size)
    W, H, bd = 256, 192, 10
    org, ref, s = make_planes(r, W, H, bd)
    if signed:
        org = (2 * org.astype(np.int32) - r.integers(0, 1024, size=org.shape)).astype(np.int16)
    cand, _ = diamond(s)
    offs1, offs2 = [], []
    for y in range(0, H, size):
        for x in range(0, W, size):
            offs1.append((PAD + y) * s + PAD + x)
            offs2.append((PAD + y + int(r.integers(-8, 9))) * s + PAD + x + int(r.integers(-8, 9)))
    jobs = D.make_jobs(offs1, offs2, dev)
    got = D.sad_jobs(torch.from_numpy(org).to(dev), s, torch.from_numpy(ref).to(dev), s, jobs, torch.from_numpy(cand).to(dev),
                     size, size, bd, signed=signed).cpu().numpy()
    d_ref = torch.from_numpy(ref).to(dev)
    dual = D.sad_jobs_dual(torch.from_numpy(org).to(dev), s, d_ref, D.plane_shift1(d_ref), s, jobs, torch.from_numpy(cand).to(dev),
                           size, size, bd, signed=signed).cpu().numpy()
    assert np.array_equal(dual, got)  # alignment-optimised form: same numbers
    O = oracle()
    pick = r.choice(len(offs1), size=min(24, len(offs1)), replace=False)
    for j in pick:
        for c in range(len(cand)):
            exp = O.xo_sad(size, size, ptr(org, offs1[j]), ptr(ref, offs2[j] + int(cand[c])), s, s, bd)
            assert got[j, c] == exp, (size, j, c)


@pytest.mark.parametrize("w,h", [(8, 8), (16, 16), (32, 32), (64, 64), (4, 4), (16, 8), (8, 32), (4, 8), (2, 2), (128, 64)])
def test_ssd_satd_diff_jobs_vs_oracle(dev, w, h):
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(200 + w * 7 + h)
    org, ref, s = make_planes(r, 256, 128, 10, pad=16)
    offs1 = [(16 + int(r.integers(0, 128 - h))) * s + 16 + int(r.integers(0, 256 - w)) for _ in range(40)]
    offs2 = [(16 + int(r.integers(0, 128 - h))) * s + 16 + int(r.integers(0, 256 - w)) for _ in range(40)]
    cand = np.array([0, 1, -1, s, -s, 3 * s + 2], np.int32)
    d_org, d_ref = torch.from_numpy(org).to(dev), torch.from_numpy(ref).to(dev)
    jobs, d_cand = D.make_jobs(offs1, offs2, dev), torch.from_numpy(cand).to(dev)
    ssd = D.ssd_jobs(d_org, s, d_ref, s, jobs, d_cand, w, h, 10).cpu().numpy()
    satd = D.satd_jobs(d_org, s, d_ref, s, jobs, d_cand, w, h, 10).cpu().numpy()
    sad = D.sad_jobs(d_org, s, d_ref, s, jobs, d_cand, w, h, 10).cpu().numpy()
    diff = D.diff_jobs(d_org, s, d_ref, s, jobs, w, h).cpu().numpy()
    O = oracle()
    for j in range(40):
        for c in range(len(cand)):
            a, b = ptr(org, offs1[j]), ptr(ref, offs2[j] + int(cand[c]))
            assert sad[j, c] == O.xo_sad(w, h, a, b, s, s, 10)
            assert ssd[j, c] == O.xo_ssd(w, h, a, b, s, s, 10)
            assert satd[j, c] == O.xo_satd(w, h, a, b, s, s, 10), (w, h, j, c)
        e = np.zeros((h, w), np.int16)
        O.xo_diff(w, h, ptr(org, offs1[j]), ptr(ref, offs2[j]), s, s, w, ptr(e))
        assert np.array_equal(diff[j], e)


@pytest.mark.parametrize("luma", [True, False])
def test_mc_jobs_vs_oracle(dev, luma):
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(300 + luma)
    O = oracle()
    pad = 72
    W, H = 192, 128
    s = W + 2 * pad
    ref = r.integers(0, 1024, size=(H + 2 * pad, s), dtype=np.int16)
    d_ref = torch.from_numpy(ref).to(dev)
    unit = 16 if luma else 32
    for (w, h) in ([(8, 8), (16, 16), (32, 32), (64, 64), (64, 16)] if luma else [(4, 4), (8, 8), (16, 16), (32, 32), (8, 32)]):
        n = 48
        gx = [(pad + int(r.integers(-40, W + 40 - w))) * unit + int(r.integers(0, unit // 4)) * 4 for _ in range(n)]
        gy = [(pad + int(r.integers(-40, H + 40 - h))) * unit + int(r.integers(0, unit // 4)) * 4 for _ in range(n)]
        frac = [int((x & (unit - 1)) != 0) | (int((y & (unit - 1)) != 0) << 1) for x, y in zip(gx, gy)]
        pred = torch.full((n, h, w), -1, dtype=torch.int16, device=dev)
        D.mc_jobs(luma, d_ref, s, pred, w, D.make_mc_jobs(gx, gy, [i * w * h for i in range(n)], frac, dev), w, h, 10)
        got = pred.cpu().numpy()
        for i in range(n):
            e = np.zeros((h, w), np.int16)
            (O.xo_mc_l if luma else O.xo_mc_c)(frac[i] & 1, frac[i] >> 1, ptr(ref), gx[i], gy[i], s, w, ptr(e), w, h, 10,
                                               O.mc_l_coeff if luma else O.mc_c_coeff)
            assert np.array_equal(got[i], e), (luma, w, h, i, gx[i] & (unit - 1), gy[i] & (unit - 1))


@pytest.mark.parametrize("lw,lh", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 5), (6, 3), (4, 6)])
def test_tq_chain_vs_oracle(dev, lw, lh):
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(400 + lw * 8 + lh)
    O = oracle()
    n, nblk, bd = 1 << (lw + lh), 24, 10
    resid = r.integers(-1023, 1024, size=(nblk, n), dtype=np.int16)
    resid[0] = 0
    resid[1] = 1023
    resid[2] = -1023
    resid[3, :] = r.integers(-3, 4, size=n)  # a block the RDOQ pre-test zeroes
    for qp, intra in ((22, 0), (32, 1), (45, 0)):
        qs, dqs = D.QUANT_SCALE[0][qp % 6], D.DQ_SCALE[qp % 6] << (qp // 6)
        d = torch.from_numpy(resid).to(dev)
        D.trans(d, lw, lh, bd)
        fwd = d.cpu().numpy().copy()
        d2 = d.clone()
        coded = D.rdoq_zero_test(d2, lw, lh, qp, qs, intra, bd).cpu().numpy()
        zt = d2.cpu().numpy()
        nnz = D.quant(d, lw, lh, qp, qs, intra, bd).cpu().numpy()
        q = d.cpu().numpy().copy()
        D.dquant(d, lw, lh, dqs, bd)
        dq = d.cpu().numpy().copy()
        D.itrans(d, lw, lh, bd)
        inv = d.cpu().numpy()
        for b in range(nblk):
            e = resid[b].copy()
            O.xo_trans(ptr(e), lw, lh, bd)
            assert np.array_equal(fwd[b], e), ("trans", b)
            c_exp = O.xo_rdoq_zero_test(ptr(e), lw, lh, qp, qs, intra, bd)
            assert coded[b] == c_exp
            assert np.array_equal(zt[b], e if c_exp else np.zeros_like(e))
            assert nnz[b] == O.xo_quant(ptr(e), lw, lh, qp, qs, intra, bd)
            assert np.array_equal(q[b], e), ("quant", b)
            O.xo_dquant(ptr(e), lw, lh, dqs, bd)
            assert np.array_equal(dq[b], e), ("dquant", b)
            O.xo_itrans(ptr(e), lw, lh, bd)
            assert np.array_equal(inv[b], e), ("itrans", b)


def test_avg_recon_vs_oracle(dev):
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(500)
    O = oracle()
    a = r.integers(0, 1024, size=4099, dtype=np.int16)
    b = r.integers(0, 1024, size=4099, dtype=np.int16)
    got = D.avg(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    e = np.zeros_like(a)
    O.xo_avg(ptr(a), ptr(b), ptr(e), 4099, 4099, 4099, 4099, 1)
    assert np.array_equal(got, e)
    nblk, w = 10, 16
    coef = r.integers(-40000, 40000, size=(nblk, w * w)).astype(np.int16)
    pred = r.integers(0, 1024, size=(nblk, w * w), dtype=np.int16)
    is_coef = np.array([i % 3 != 0 for i in range(nblk)], np.uint8)
    s_rec = 200
    rec_off = np.array([i * w for i in range(nblk)], np.int32)
    rec = torch.zeros((w, s_rec), dtype=torch.int16, device=dev)
    D.recon(torch.from_numpy(coef).to(dev), torch.from_numpy(pred).to(dev), torch.from_numpy(is_coef).to(dev), w, w,
            torch.from_numpy(rec_off).to(dev), s_rec, rec, 10)
    got = rec.cpu().numpy()
    for i in range(nblk):
        e = np.zeros((w, s_rec), np.int16)
        O.xo_recon(ptr(coef, i * w * w), ptr(pred, i * w * w), int(is_coef[i]), w, w, s_rec, ptr(e), 10)
        assert np.array_equal(got[:, i * w:(i + 1) * w], e[:, :w])


# ---------------------------------------------------------------------------------------------------------
# full-size (3840x2160) properties: no oracle at this size, identities of the arithmetic instead
# ---------------------------------------------------------------------------------------------------------
def test_full_size_properties_4k(dev):
    import torch

    from xeve_amd import device as D

    W, H, bd = 3840, 2160, 10
    s = W + 2 * PAD
    g = torch.Generator(device=dev).manual_seed(4)
    org = torch.randint(0, 1024, (H + 2 * PAD, s), generator=g, device=dev, dtype=torch.int16)
    ref = torch.randint(0, 1024, (H + 2 * PAD, s), generator=g, device=dev, dtype=torch.int16)
    ys, xs = np.meshgrid(np.arange(0, H - 63, 64), np.arange(0, W, 64), indexing="ij")
    off64 = ((PAD + ys) * s + PAD + xs).ravel().astype(np.int32)
    cand = torch.tensor([0, 1, -s, 5 * s - 3, -64 * s + 64], dtype=torch.int32, device=dev)
    jobs64 = D.make_jobs(off64, off64, dev)
    # (1) a block against itself has zero distortion, for every size class
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    for size in (8, 16, 32, 64):
        sad0 = D.sad_jobs(org, s, org, s, jobs64, z, size, size, bd)
        assert int(sad0.abs().sum()) == 0
        assert int(D.satd_jobs(org, s, org, s, jobs64, z, size, size, bd).abs().sum()) == 0
        assert int(D.ssd_jobs(org, s, org, s, jobs64, z, size, size, bd).abs().sum()) == 0
    # (2) SAD is additive over a quad-tree split BEFORE the final shift: at bit depth 8 (shift 0) the 64x64 SAD
    #     equals the sum of its four 32x32 SADs, which equal the sums of their 16x16 and 8x8 SADs  
    sad64 = D.sad_jobs(org, s, ref, s, jobs64, cand, 64, 64, 8).to(torch.int64)
    for size in (32, 16, 8):
        k = 64 // size
        sub = (off64[:, None, None] + (np.arange(k)[None, :, None] * size * s) + np.arange(k)[None, None, :] * size).reshape(-1)
        jobs = D.make_jobs(sub, sub, dev)
        part = D.sad_jobs(org, s, ref, s, jobs, cand, size, size, 8).to(torch.int64)
        assert torch.equal(part.view(len(off64), k * k, -1).sum(1), sad64), size
    # (3) SSD at bit depth 8 equals the sum of squares of DIFF; recon(diff, pred) returns the original
    ssd64 = D.ssd_jobs(org, s, ref, s, jobs64, z, 64, 64, 8)
    diff = D.diff_jobs(org, s, ref, s, jobs64, 64, 64)
    assert torch.equal((diff.to(torch.int64) ** 2).sum((1, 2)), ssd64[:, 0])
    pred = torch.empty((len(off64), 64, 64), dtype=torch.int16, device=dev)
    mcj = D.make_mc_jobs(((off64 % s)) * 16, (off64 // s) * 16, np.arange(len(off64)) * 4096, np.zeros(len(off64)), dev)
    D.mc_jobs(True, ref, s, pred, 64, mcj, 64, 64, bd)  # integer-pel MC is a copy
    rec = torch.zeros_like(org)
    D.recon(diff.view(len(off64), -1), pred.view(len(off64), -1), None, 64, 64, torch.from_numpy(off64).to(dev), s, rec, bd)
    tiles = lambda p: torch.stack([p[PAD + y:PAD + y + 64, PAD + x:PAD + x + 64] for y, x in zip(ys.ravel()[:50], xs.ravel()[:50])])
    assert torch.equal(tiles(rec), tiles(org))
    assert torch.equal(pred[:50], tiles(ref))
    # (4) transform linearity / structure on all 64x64 residual tiles of the picture: T(0) = 0, the 64-point
    #     transform only populates the 32x32 low-frequency corner, and T(-x) = -T(x) up to the rounding offset
    x = diff.view(len(off64), -1).clone()
    D.trans(x, 6, 6, bd)
    c = x.view(-1, 64, 64)
    assert int(c[:, 32:, :].abs().sum()) == 0 and int(c[:, :, 32:].abs().sum()) == 0 and int(c[:, :32, :32].abs().sum()) > 0
    xn = (-diff).view(len(off64), -1).clone()
    D.trans(xn, 6, 6, bd)
    assert int((x.to(torch.int32) + xn.to(torch.int32)).abs().max()) <= 1
    zero = torch.zeros((4, 4096), dtype=torch.int16, device=dev)
    assert int(D.trans(zero, 6, 6, bd).abs().sum()) == 0
    # (5) avg(a, a) = a over the whole padded plane
    assert torch.equal(D.avg(org.view(-1), org.view(-1)), org.view(-1))
    torch.cuda.synchronize()


@pytest.mark.parametrize("lg", [5, 6])
def test_dct_matrix_core_path_full_int16_range(dev, lg):
    """32x32 / 64x64 run on v_mfma_i32_32x32x32_i8 with byte-split operands (xeve_amd/csrc/dct_mfma.hip): exact for
    EVERY int16 input, including the extremes that stress the byte split and the offset-binary correction terms."""
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(600 + lg)
    O = oracle()
    n = 1 << (2 * lg)
    x = r.integers(-32768, 32768, size=(12, n), dtype=np.int16)
    x[0], x[1], x[2], x[3] = 32767, -32768, 0, 1
    x[4, ::2], x[4, 1::2] = 32767, -32768
    x[5] = (np.arange(n) % 251 - 125) * 262  # smooth-ish, asymmetric: catches transposed layouts
    d = torch.from_numpy(x).to(dev)
    D.trans(d, lg, lg, 10)
    got = d.cpu().numpy()
    for b in range(12):
        e = x[b].copy()
        O.xo_trans(ptr(e), lg, lg, 10)
        assert np.array_equal(got[b], e), ("fwd", lg, b)
    d = torch.from_numpy(x).to(dev)
    D.itrans(d, lg, lg, 10)
    got = d.cpu().numpy()
    for b in range(12):
        e = x[b].copy()
        O.xo_itrans(ptr(e), lg, lg, 10)
        assert np.array_equal(got[b], e), ("inv", lg, b)


@pytest.mark.parametrize("lw,lh", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (2, 4), (5, 3), (6, 5)])
def test_fused_residual_rdo_vs_oracle(dev, lw, lh):
    """xeve_hip_residual_rdo = DIFF, SSD, DCT, zero pre-test, quant, dequant, IDCT, recon, SSD in one launch
    (matrix cores for 32x32 / 64x64) against the oracle's step-by-step chain."""
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(700 + lw * 8 + lh)
    O = oracle()
    w, h, bd = 1 << lw, 1 << lh, 10
    n, nblk = w * h, 21
    s = 4 * 64 + 32
    org = r.integers(0, 1024, size=(6 * 64 + 16, s), dtype=np.int16)
    pred = r.integers(0, 1024, size=(nblk, n), dtype=np.int16)
    offs = [(8 + (b // 4) * h) * s + 8 + (b % 4) * w for b in range(nblk)]
    for b in (0, 1, 2):  # near-perfect predictions: exercise the all-zero pre-test and small levels
        blk = org.reshape(-1)[np.add.outer(np.arange(h) * s, np.arange(w)).ravel() + offs[b]]
        pred[b] = np.clip(blk + r.integers(-b - 1, b + 2, size=n), 0, 1023)
    for qp, intra, zt in ((27, 0, 1), (37, 1, 1), (32, 0, 0)):
        d_org, d_pred = torch.from_numpy(org).to(dev), torch.from_numpy(pred).to(dev)
        jobs = D.make_jobs(offs, np.arange(nblk) * n, dev)
        coef = torch.full((nblk, n), 77, dtype=torch.int16, device=dev)
        rec = torch.zeros_like(d_org)
        nnz = torch.zeros(nblk, dtype=torch.int32, device=dev)
        ssd = torch.zeros((nblk, 2), dtype=torch.int64, device=dev)
        D.residual_rdo(d_org, s, d_pred, w, jobs, lw, lh, bd, qp, intra, zt, coef, rec, s, nnz, ssd)
        coef, rec, nnz, ssd = coef.cpu().numpy(), rec.cpu().numpy(), nnz.cpu().numpy(), ssd.cpu().numpy()
        qs, dqs = D.QUANT_SCALE[0][qp % 6], D.DQ_SCALE[qp % 6] << (qp // 6)
        for b in range(nblk):
            c = np.zeros(n, np.int16)
            O.xo_diff(w, h, ptr(org, offs[b]), ptr(pred, b * n), s, w, w, ptr(c))
            assert ssd[b, 0] == O.xo_ssd(w, h, ptr(org, offs[b]), ptr(pred, b * n), s, w, bd)
            O.xo_trans(ptr(c), lw, lh, bd)
            if zt and not O.xo_rdoq_zero_test(ptr(c), lw, lh, qp, qs, intra, bd):
                c[:] = 0
                e_nnz = 0
            else:
                e_nnz = O.xo_quant(ptr(c), lw, lh, qp, qs, intra, bd)
            assert nnz[b] == e_nnz and np.array_equal(coef[b], c), ("levels", lw, lh, qp, b)
            O.xo_dquant(ptr(c), lw, lh, dqs, bd)
            O.xo_itrans(ptr(c), lw, lh, bd)
            e = np.zeros((h, s), np.int16)
            O.xo_recon(ptr(c), ptr(pred, b * n), 1, w, h, s, ptr(e), bd)
            y0, x0 = offs[b] // s, offs[b] % s
            assert np.array_equal(rec[y0:y0 + h, x0:x0 + w], e[:, :w]), ("rec", lw, lh, qp, b)
            assert ssd[b, 1] == O.xo_ssd(w, h, ptr(org, offs[b]), ptr(rec, offs[b]), s, s, bd)


@pytest.mark.parametrize("luma", [True, False])
def test_fused_mc_distortion_vs_oracle(dev, luma):
    """xeve_hip_mc_l_sad_jobs / xeve_hip_mc_ssd_jobs: interpolate and compare with the original in one launch."""
    import torch

    from xeve_amd import device as D

    r = np.random.default_rng(800 + luma)
    O = oracle()
    pad, W, H = 72, 192, 128
    s = W + 2 * pad
    ref = r.integers(0, 1024, size=(H + 2 * pad, s), dtype=np.int16)
    org = r.integers(0, 1024, size=(H + 2 * pad, s), dtype=np.int16)
    d_ref, d_org = torch.from_numpy(ref).to(dev), torch.from_numpy(org).to(dev)
    unit = 16 if luma else 32
    for (w, h) in ([(8, 8), (16, 16), (32, 32), (64, 64)] if luma else [(4, 4), (8, 8), (16, 16), (32, 32)]):
        n = 70
        gx = [(pad + int(r.integers(-30, W + 30 - w))) * unit + int(r.integers(0, unit // 4)) * 4 for _ in range(n)]
        gy = [(pad + int(r.integers(-30, H + 30 - h))) * unit + int(r.integers(0, unit // 4)) * 4 for _ in range(n)]
        frac = [int((x & (unit - 1)) != 0) | (int((y & (unit - 1)) != 0) << 1) for x, y in zip(gx, gy)]
        ooff = [(pad + int(r.integers(0, H - h))) * s + pad + int(r.integers(0, W - w)) for _ in range(n)]
        jobs = D.make_mc_jobs(gx, gy, ooff, frac, dev)
        ssd = D.mc_ssd_jobs(luma, d_ref, s, d_org, s, jobs, w, h, 10, torch.zeros(n, dtype=torch.int64, device=dev)).cpu().numpy()
        sad = D.mc_l_sad_jobs(d_ref, s, d_org, s, jobs, w, h, 10, torch.zeros(n, dtype=torch.int32, device=dev)).cpu().numpy() if luma else None
        for i in range(n):
            e = np.zeros((h, w), np.int16)
            (O.xo_mc_l if luma else O.xo_mc_c)(frac[i] & 1, frac[i] >> 1, ptr(ref), gx[i], gy[i], s, w, ptr(e), w, h, 10,
                                               O.mc_l_coeff if luma else O.mc_c_coeff)
            assert ssd[i] == O.xo_ssd(w, h, ptr(org, ooff[i]), ptr(e), s, w, 10), (luma, w, i)
            if luma:
                assert sad[i] == O.xo_sad(w, h, ptr(org, ooff[i]), ptr(e), s, w, 10), (w, i)
