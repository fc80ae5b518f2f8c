"""Seeded cases for pinter_residue_rdo: an original picture that is (mostly) a displaced, noisy copy of the reference pictures, so that
candidates range from "predicts perfectly, nothing to code" to "random vector, dense residual"."""
import numpy as np

from _libs import RDO_JOB_DTYPE, RdoParams
from _mc_cases import PAD_L, make_refs
from _sbac_cases import make_states


def make_picture(r, w, h, bd, nref, idc=1):
    refs = make_refs(r, w, h, bd, nref, idc)
    maxv = (1 << bd) - 1
    # smooth the reference pictures a little (pure noise never quantises to zero) and derive the original from reference (0, list 0)
    for pic in refs["pics"]:
        for k in range(3):
            a = pic[k].astype(np.int32)
            a = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, (1, 1), (0, 1)) + 2) >> 2
            a = (a + np.roll(a, 2, 0) + np.roll(a, 2, 1) + np.roll(a, (2, 2), (0, 1)) + 2) >> 2
            pic[k][:] = a.astype(np.int16)
    amp = int(r.choice([1, 3, 12]))
    org = [np.clip(p.astype(np.int32) + r.integers(-amp, amp + 1, size=p.shape), 0, maxv).astype(np.int16) for p in refs["pics"][0]]
    return refs, org


def make_params(r, lw, lh, w, h, bd, nref, idc=1, slice_type=0):
    p = RdoParams()
    p.log2_cuw, p.log2_cuh, p.pic_w, p.pic_h, p.slice_type, p.chroma_format_idc, p.bit_depth, p.tool_iqt = lw, lh, w, h, slice_type, idc, bd, 0
    p.num_refp[0], p.num_refp[1] = nref, (nref if slice_type == 0 else 0)
    qp = int(r.integers(22, 46)) + 6 * (bd - 8)
    dq = int(r.integers(-3, 4))
    p.qp[0], p.qp[1], p.qp[2] = qp, max(0, qp + dq), max(0, qp + dq - 1)
    lam = 0.57 * 2.0 ** ((qp - 6 * (bd - 8) - 12) / 3.0) * (0.8 + 0.4 * float(r.random()))
    p.lambda_[0] = lam
    p.dist_chroma_weight[0], p.dist_chroma_weight[1] = 2.0 ** (-dq / 3.0), 2.0 ** ((1 - dq) / 3.0)
    p.lambda_[1], p.lambda_[2] = lam / p.dist_chroma_weight[0], lam / p.dist_chroma_weight[1]
    return p


def make_jobs(r, n, w, h, cuw, cuh, nref, nstates, slice_type=0):
    j = np.zeros(n, RDO_JOB_DTYPE)
    j["x"] = r.integers(0, max(1, (w - cuw) // 4 + 1), size=n) * 4
    j["y"] = r.integers(0, max(1, (h - cuh) // 4 + 1), size=n) * 4
    kind = r.integers(0, 3, size=n) if slice_type == 0 else np.zeros(n, np.int64)  # 0 L0, 1 L1, 2 BI
    good = r.random(n) < 0.6  # near the true motion (0, 0) of reference (0, list 0)
    mv = np.where(good[:, None, None], r.integers(-2, 3, size=(n, 2, 2)), r.integers(-60, 61, size=(n, 2, 2)))
    j["mv"] = mv
    j["mvd"] = r.choice([0, 0, 1, -1, 3, -6, 17, -40, 300], size=(n, 2, 2))
    j["refi"][:, 0] = np.where(kind == 1, -1, np.where(good, 0, r.integers(0, nref, size=n)))
    j["refi"][:, 1] = np.where(kind == 0, -1, r.integers(0, nref, size=n))
    j["mvp_idx"] = r.integers(0, 4, size=(n, 2))
    j["dir_flag"] = r.random(n) < 0.15
    j["ctx_skip"], j["ctx_pred_mode"] = r.integers(0, 2, size=n), r.integers(0, 3, size=n)
    j["sbac"] = r.integers(0, nstates, size=n)
    return j


def states(r, n):
    return make_states(r, n)


def make_skip_jobs(r, n, w, h, cuw, cuh, nstates, ncand):
    """merge candidates as xeve_get_motion derives them in Baseline: four vectors per list, reference index 0; duplicates (pruned by the
    reference), the (1, 1) vector of an unavailable neighbour, near and far vectors"""
    from _libs import SKIP_JOB_DTYPE

    j = np.zeros(n, SKIP_JOB_DTYPE)
    j["x"] = r.integers(0, max(1, (w - cuw) // 4 + 1), size=n) * 4
    j["y"] = r.integers(0, max(1, (h - cuh) // 4 + 1), size=n) * 4
    mv = np.where(r.random((n, 2, 4, 1)) < 0.7, r.integers(-6, 7, size=(n, 2, 4, 2)), r.integers(-300, 301, size=(n, 2, 4, 2)))
    mv[r.random((n, 2, 4)) < 0.15] = 1
    dup = r.random((n, 2)) < 0.4
    for l in range(2):
        mv[dup[:, l], l, 1] = mv[dup[:, l], l, 0]
    j["mvp"] = mv
    j["refi_pred"] = 0
    j["ncand"] = ncand
    j["sbac"] = r.integers(0, nstates, size=n)
    j["ctx_skip"] = r.integers(0, 2, size=n)
    return j


def fuzz_cases(n_iter, seed0=0, n_jobs=40, n_skip=30):
    """random configurations for pinter_residue_rdo and xeve_analyze_skip: bit depth 8 / 10 / 12, chroma format, slice type, 1-3 reference pictures, square
    and non-square CUs, and per size one of: plain, QP / lambda extremes, vectors far outside the picture with the CU on a picture corner.
    Yields (refs, org, states, p, lw, lh, rdo_jobs, skip_jobs or None, ncand, meta)."""
    for it in range(n_iter):
        r = np.random.default_rng(20_000 + seed0 + it)
        w, h = int(r.choice([64, 128, 192])), int(r.choice([64, 128]))
        bd, idc, st_type, nref = int(r.choice([8, 10, 10, 12])), int(r.choice([0, 1, 1, 3])), int(r.choice([0, 0, 1])), int(r.choice([1, 2, 3]))
        refs, org = make_picture(r, w, h, bd, nref, idc)
        st = states(r, 6)
        for (lw, lh) in [(3, 3), (4, 4), (5, 5), (6, 6), (2, 2), (4, 3), (3, 5), (6, 4)]:
            cuw, cuh = 1 << lw, 1 << lh
            if cuw > w or cuh > h:
                continue
            p = make_params(r, lw, lh, w, h, bd, nref, idc, st_type)
            kind = int(r.integers(0, 3))
            if kind == 1:
                q = int(r.choice([0, 4, 51 + 6 * (bd - 8)]))
                p.qp[0] = p.qp[1] = p.qp[2] = q
                lam = float(r.choice([1e-3, 0.5, 5e4]))
                p.lambda_[0] = p.lambda_[1] = p.lambda_[2] = lam
            jobs = make_jobs(r, n_jobs, w, h, cuw, cuh, nref, len(st), st_type)
            if kind == 2:
                jobs["mv"] = r.integers(-1500, 1501, size=jobs["mv"].shape)
                jobs["x"], jobs["y"] = r.choice([0, w - cuw], size=len(jobs)), r.choice([0, h - cuh], size=len(jobs))
                jobs["mvd"] = r.integers(-4000, 4001, size=jobs["mvd"].shape)
            sj, ncand = None, 0
            if lw == lh:
                ncand = int(r.integers(1, 5))
                sj = make_skip_jobs(r, n_skip, w, h, cuw, cuh, len(st), ncand)
                sj["refi_pred"] = r.integers(-1, nref, size=sj["refi_pred"].shape)
                if kind == 2:
                    sj["mvp"] = r.integers(-1500, 1501, size=sj["mvp"].shape)
                    sj["x"], sj["y"] = r.choice([0, w - cuw], size=len(sj)), r.choice([0, h - cuh], size=len(sj))
            yield refs, org, st, p, lw, lh, jobs, sj, ncand, dict(it=it, lw=lw, lh=lh, kind=kind, w=w, h=h, bd=bd, idc=idc, slice_type=st_type, nref=nref)
