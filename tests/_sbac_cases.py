"""Seeded cases for the CABAC bit counting of an inter CU (xeve_rdo_bit_cnt_cu_inter & co., xeve_mode.c:39-295)."""
import numpy as np

from _libs import CU_BITS_JOB_DTYPE, SBAC_DTYPE, SBAC_NCTX, CuBitsParams


def make_states(r, n):
    """plausible coder states: range in its renormalised interval, any code, every model a valid (state, mps) pair"""
    s = np.zeros(n, SBAC_DTYPE)
    s["range"] = r.integers(8192, 16385, size=n)
    s["code"] = r.integers(0, 1 << 25, size=n)
    s["code_bits"] = r.integers(1, 12, size=n)  # xeve_sbac_bit_reset overwrites these; the values must not matter
    s["stacked_ff"] = r.integers(0, 3, size=n)
    s["bitcounter"] = r.integers(0, 1000, size=n)
    # the 68 models of the inter-CU syntax come from `r` exactly as when the committed goldens were made (their inputs are regenerated from the seeds);
    # the models added since (intra_dir, split_cu_flag, delta_qp: indices 68..71) from a generator of their own, so the stream of `r` is unchanged
    s["ctx"][:, :68] = (r.integers(1, 257, size=(n, 68)) << 1) | r.integers(0, 2, size=(n, 68))
    r2 = np.random.default_rng(int(s["code"][0]) + 68)
    s["ctx"][:, 68:] = (r2.integers(1, 257, size=(n, SBAC_NCTX - 68)) << 1) | r2.integers(0, 2, size=(n, SBAC_NCTX - 68))
    s["ctx"][0] = 512  # PROB_INIT everywhere: the state at the start of a slice
    return s


def make_levels(r, n, kind):
    """quantised levels of one block (dense raster order)"""
    if kind == 0:  # dense, large levels (what i.i.d. synthetic pictures give)
        c = r.integers(-40, 41, size=n)
    elif kind == 1:  # sparse, small levels (typical inter residual)
        c = np.where(r.random(n) < 0.08, r.integers(-3, 4, size=n), 0)
    elif kind == 2:  # a few isolated big levels, long zero runs
        c = np.zeros(n, np.int64)
        k = max(1, n // 64)
        c[r.integers(0, n, size=k)] = r.integers(-2000, 2001, size=k)
    elif kind == 3:  # extremes of the s16 range (a few: each costs 32768 bins), last position occupied
        c = np.where(r.random(n) < 0.3, r.choice([1, -1, 2, -7], size=n), 0)
        c[r.integers(0, n, size=3)] = r.choice([-32768, 32767, 32766], size=3)
        c[-1] = -32768
    else:  # DC only
        c = np.zeros(n, np.int64)
        c[0] = int(r.integers(1, 9))
    return c.astype(np.int16)


def make_jobs(r, njobs, lw, lh, nstates, idc=1, nnz_mode=0):
    """jobs + their coefficient buffer.  nnz_mode 0: exact counts; 1: some counts too small / too large / forced zero"""
    ws, hs = (1 if idc <= 2 else 0), (1 if idc <= 1 else 0)
    ny, nc = 1 << (lw + lh), 1 << (lw + lh - ws - hs)
    coef = np.zeros(njobs * (ny + 2 * nc), np.int16)
    jobs = np.zeros(njobs, CU_BITS_JOB_DTYPE)
    for i in range(njobs):
        base = i * (ny + 2 * nc)
        offs = (base, base + ny, base + ny + nc)
        for c, (o, n) in enumerate(zip(offs, (ny, nc, nc))):
            if idc == 0 and c:
                blk = np.zeros(n, np.int16)
            else:
                blk = make_levels(r, n, int(r.integers(0, 5))) if r.random() < 0.85 else np.zeros(n, np.int16)
            coef[o:o + n] = blk
            nnz = int(np.count_nonzero(blk))
            if nnz_mode and nnz:
                u = r.random()
                nnz = 0 if u < 0.2 else (max(1, nnz - 1) if u < 0.35 else (nnz + 2 if u < 0.5 else nnz))
            jobs["nnz"][i, c] = nnz
        jobs["coef_off"][i] = offs
        jobs["sbac"][i] = int(r.integers(0, nstates))
        jobs["mvd"][i] = r.choice([0, 0, 1, -1, 2, -3, 7, -8, 15, 16, -100, 511, -2047, 4095, -32768, 32767], size=(2, 2))
        pd = int(r.integers(0, 3))  # L0 / L1 / BI
        jobs["refi"][i] = [(int(r.integers(0, 3)) if pd != 1 else -1), (int(r.integers(0, 3)) if pd != 0 else -1)]
        jobs["mvp_idx"][i] = r.integers(0, 4, size=2)
        jobs["mode"][i] = int(r.integers(0, 5))
        jobs["dir_flag"][i] = int(r.random() < 0.2)
        jobs["ctx_skip"][i] = int(r.integers(0, 2))
        jobs["ctx_pred_mode"][i] = int(r.integers(0, 3))
    return jobs, coef


def make_params(lw, lh, slice_type=0, num_refp=(2, 2), cm_init=0, idc=1):
    p = CuBitsParams()
    p.log2_cuw, p.log2_cuh, p.slice_type, p.cm_init, p.chroma_format_idc = lw, lh, slice_type, cm_init, idc
    p.num_refp[0], p.num_refp[1] = num_refp
    return p


def clamp_refi(jobs, num_refp):
    """reference indices must exist in their list"""
    for l in range(2):
        v = jobs["refi"][:, l]
        jobs["refi"][:, l] = np.where(v >= 0, np.minimum(v, max(num_refp[l] - 1, 0)), v)
    return jobs
