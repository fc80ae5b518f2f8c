"""Seeded cases for the whole inter analysis of a CU (xeve_pinter_analyze_cu), shared by the reference-pinning test, the golden generator and the
GPU tests: a smooth texture seen through differently shifted, noisy reference pictures, so that the motion searches walk, the reference pictures
compete and all of skip / direct / L0 / L1 / bi win somewhere."""
import numpy as np

from _libs import INTER_JOB_DTYPE, InterParams
from _mc_cases import make_refs
from _rdo_cases import make_params


def make_inter_picture(r, w, h, bd, nref, idc=1, slice_type=0, gop=8):
    refs = make_refs(r, w, h, bd, nref, idc)
    maxv, ws, hs = (1 << bd) - 1, refs["ws"], refs["hs"]
    amp = maxv / 1023.0
    noise = int(r.choice([2, 6, 14]))

    def texture(shape, scale):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
        yy, xx = yy * scale, xx * scale
        t = 512 + 300 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 150 * np.sin((xx + yy) / 41.0) + 60 * np.sin(xx / 5.0 + yy / 7.0)
        return t * amp

    base = [texture(refs["pics"][0][0].shape, 1.0), texture(refs["pics"][0][1].shape, 1.0 + ws), texture(refs["pics"][0][2].shape, 1.0 + ws) * 0.7 + 150 * amp]
    shifts = []
    for i, pic in enumerate(refs["pics"]):
        sy, sx = int(r.integers(-5, 6)), int(r.integers(-5, 6))
        shifts.append((sx * 2, sy * 2))
        for k in range(3):
            sh = (sy * 2, sx * 2) if k == 0 else (sy * (2 >> hs), sx * (2 >> ws))
            a = np.roll(base[k], sh, axis=(0, 1)) + r.integers(-noise, noise + 1, size=base[k].shape) * amp
            pic[k][:] = np.clip(a, 0, maxv).astype(np.int16)
    org = [np.clip(b + r.integers(-noise, noise + 1, size=b.shape) * amp, 0, maxv).astype(np.int16) for b in base]
    # POCs: list 0 in the past, list 1 in the future (B) -- the table is indexed [refi * 2 + list]
    poc = 16
    for i in range(nref):
        refs["pocs"][2 * i + 0] = poc - (i + 1) * (gop // 2 if slice_type == 0 else 1)
        refs["pocs"][2 * i + 1] = poc + (i + 1) * (gop // 2) if slice_type == 0 else poc - (i + 1)
    refs["poc"], refs["gop"], refs["shifts"] = poc, gop, shifts
    return refs, org


def refi_bits(num_refp, refi):
    """xeve_tbl_refi_bits[num_refp][refi] (xeve_tbl.c:498-517) in closed form"""
    return 0 if num_refp < 2 else min(refi + 1, num_refp - 1)


def make_inter_params(r, lw, w, h, bd, nref, idc, slice_type, refs, skip_th=0.0, max_cand=None, nref1=None):
    P = InterParams()
    rp = make_params(r, lw, lw, w, h, bd, nref, idc, slice_type)
    if nref1 is not None and slice_type == 0:
        rp.num_refp[1] = nref1  # list 1 shorter than list 0 (analyze_bi then walks both lists with list 1's count)
    P.rdo = rp
    msr = int(r.choice([32, 64]))
    lam_mv = int(np.floor(65536.0 * np.sqrt(rp.lambda_[0])))
    P.me.lambda_mv, P.me.max_search_range, P.me.faststep = lam_mv, msr, 3
    P.me.min_clip[0], P.me.min_clip[1], P.me.max_clip[0], P.me.max_clip[1] = -127, -127, w - 1, h - 1  # -MAX_CU_SIZE + 1 .. pic - 1 (xeve_pinter.c:2124-2127)
    P.spel.lambda_mv, P.spel.hpel_cnt, P.spel.qpel_cnt = lam_mv, int(r.choice([4, 8])), int(r.choice([0, 8, 8]))
    for l in range(2):
        for i in range(nref):
            P.refi_bits[l][i] = refi_bits(rp.num_refp[l], i)
            d = abs(refs["poc"] - int(refs["pocs"][2 * i + l]))
            P.range_recentre[l][i] = min(max((msr * d + (refs["gop"] >> 1)) // refs["gop"], msr >> 2), msr)  # get_range_ipel (xeve_pinter.c:124-129)
    P.max_cand = int(r.choice([2, 3, 4])) if max_cand is None else max_cand
    P.poc, P.col_list_poc0, P.skip_th = refs["poc"], refs["poc"] - int(r.choice([0, 4, 8, 12])), skip_th
    return P


def make_inter_jobs(r, n, w, h, cu, nstates, refs, slice_type):
    j = np.zeros(n, INTER_JOB_DTYPE)
    j["x"] = r.integers(0, (w - cu) // 8 + 1, size=n) * 8
    j["y"] = r.integers(0, (h - cu) // 8 + 1, size=n) * 8
    mv = np.zeros((n, 2, 4, 2), np.int64)
    for l in range(2):
        sx, sy = refs["shifts"][0 * 2 + l]  # the true motion towards reference 0 of the list (quarter pel: 4 * shift)
        true = np.array([4 * sx, 4 * sy])
        kind = r.integers(0, 4, size=(n, 4))
        mv[:, l] = np.where(kind[..., None] == 0, true + r.integers(-3, 4, size=(n, 4, 2)),
                            np.where(kind[..., None] == 1, r.integers(-60, 61, size=(n, 4, 2)), np.where(kind[..., None] == 2, 1, true)))
    dup = r.random((n, 2)) < 0.3
    for l in range(2):
        mv[dup[:, l], l, 1] = mv[dup[:, l], l, 0]
    j["mvp"] = mv
    j["mv_col"] = r.integers(-40, 41, size=(n, 2))
    j["sbac"] = r.integers(0, nstates, size=n)
    j["ctx_skip"] = r.integers(0, 2, size=n)
    j["ctx_pred_mode"] = r.integers(0, 3, size=n)
    return j


def mask_unobservable(res, slice_type):
    """fields the reference leaves stale: motion data of a list the winning mode does not use; candidate indices of the direct mode"""
    res = res.copy()
    for l in range(2):
        off = res["refi"][:, l] < 0
        res["mv"][off, l], res["mvd"][off, l], res["mvp_idx"][off, l] = 0, 0, 0
    d = res["cu_mode"] == 3
    res["mvp_idx"][d] = 0
    res["cost_inter"], res["best_idx"] = 0, 0
    return res


def make_maps(r, w_scu, h_scu, tiles=1):
    """per-4x4-unit maps as the encoder keeps them: map_scu (bit 15 intra, 26 IBC, 31 coded), map_tidx, map_mv and the two collocated maps"""
    n = w_scu * h_scu
    coded = r.random(n) < 0.8
    intra = r.random(n) < 0.2
    ibc = r.random(n) < 0.05
    junk = r.integers(0, 1 << 15, size=n).astype(np.uint32)  # qp / depth bits below
    map_scu = (junk | (intra.astype(np.uint32) << 15) | (ibc.astype(np.uint32) << 26) | (coded.astype(np.uint32) << 31)).astype(np.uint32)
    tidx = np.zeros(n, np.uint8) if tiles == 1 else (np.arange(n) % w_scu >= w_scu // 2).astype(np.uint8)
    mk = lambda: r.integers(-300, 301, size=(n, 2, 2)).astype(np.int16)
    return map_scu, tidx, mk(), mk(), mk()


def oracle_params_from_hip(hp):
    """lib.InterParams (the library's layout) -> _libs.InterParams (the oracle's: xo_epzs_params carries a separate sub-pel record)"""
    import ctypes as C

    P = InterParams()
    C.memmove(C.byref(P.rdo), C.byref(hp.rdo), C.sizeof(P.rdo))
    C.memmove(C.byref(P.me), C.byref(hp.me.me), C.sizeof(P.me))
    P.spel.lambda_mv, P.spel.hpel_cnt, P.spel.qpel_cnt = hp.me.me.lambda_mv, hp.me.hpel_cnt, hp.me.qpel_cnt
    for l in range(2):
        for i in range(8):
            P.refi_bits[l][i], P.range_recentre[l][i] = hp.refi_bits[l][i], hp.range_recentre[l][i]
    P.max_cand, P.poc, P.col_list_poc0, P.skip_th = hp.max_cand, hp.poc, hp.col_list_poc0, hp.skip_th
    return P


def fuzz_cases(n_iter, seed0=0, n_jobs=36):
    """random configurations for the whole inter analysis: picture size, bit depth 8 / 10 / 12, chroma format, slice type, 1-3 reference pictures
    (list 1 possibly shorter), skip_th, and per level one of: plain, QP / lambda extremes, candidates far outside the picture with the CU on a
    picture corner, all candidates equal / zero.  Yields (refs, org, states, P, jobs, meta)."""
    from _rdo_cases import states

    for it in range(n_iter):
        r = np.random.default_rng(10_000 + seed0 + it)
        w, h = int(r.choice([64, 128, 192])), int(r.choice([64, 128]))
        bd, idc, st_type = int(r.choice([8, 10, 10, 12])), int(r.choice([0, 1, 1, 1, 3])), int(r.choice([0, 0, 1]))
        nref = int(r.choice([1, 2, 3]))
        refs, org = make_inter_picture(r, w, h, bd, nref, idc, st_type)
        st = states(r, 6)
        for lw in (3, 4, 5, 6):
            cu = 1 << lw
            P = make_inter_params(r, lw, w, h, bd, nref, idc, st_type, refs, float(r.choice([0.0, 0.0, 3.0, 50.0])),
                                  nref1=(int(r.integers(1, nref + 1)) if st_type == 0 else None))
            if r.random() < 0.3:  # me_complexity > 1: me_raster after a first search that ended far from its start
                P.me.reserved = 1
            if r.random() < 0.25:  # me_level = ME_LEV_IPEL: integer refinement instead of the sub-pel pattern
                P.spel.hpel_cnt = P.spel.qpel_cnt = 0
            kind = int(r.integers(0, 4))
            if kind == 1:  # extreme QP / lambda
                q = int(r.choice([0, 4, 51 + 6 * (bd - 8)]))
                P.rdo.qp[0] = P.rdo.qp[1] = P.rdo.qp[2] = q
                lam = float(r.choice([1e-3, 0.5, 5e4]))
                P.rdo.lambda_[0] = P.rdo.lambda_[1] = P.rdo.lambda_[2] = lam
                P.me.lambda_mv = P.spel.lambda_mv = int(np.floor(65536.0 * np.sqrt(lam))) & 0xFFFFFFFF
            jobs = make_inter_jobs(r, n_jobs, w, h, cu, len(st), refs, st_type)
            if kind == 2:  # candidates far outside the picture, CUs on the corners
                jobs["mvp"] = r.integers(-1200, 1201, size=jobs["mvp"].shape)
                jobs["x"] = r.choice([0, w - cu], size=len(jobs))
                jobs["y"] = r.choice([0, h - cu], size=len(jobs))
                jobs["mv_col"] = r.integers(-2000, 2001, size=jobs["mv_col"].shape)
            if kind == 3:  # all candidates equal (everything pruned but the first), zero vectors
                jobs["mvp"][:] = jobs["mvp"][:, :, :1]
                jobs["mvp"][: len(jobs) // 2] = 0
            yield refs, org, st, P, jobs, dict(it=it, lw=lw, kind=kind, w=w, h=h, bd=bd, idc=idc, slice_type=st_type, nref=nref)
