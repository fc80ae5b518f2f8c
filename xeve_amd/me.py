"""Per-list motion search of one picture level on the GPU: the host-side loop of pinter_me_epzs
(reference: src_base/xeve_pinter.c:699-869, me_complexity 1 -- no raster search -- and me_level > ME_LEV_IPEL) over the
two device searches xeve_hip_me_ipel_diamond_jobs and xeve_hip_me_spel_pattern_jobs.

The reference runs this loop per CU on the CPU; here every step is one launch over ALL blocks of a quad-tree level, and
the only host work is the bookkeeping the reference does between the calls (compare costs, re-centre the range, decide
which blocks still refine), vectorised over the blocks in numpy.  No pixel arithmetic happens on the host.
"""
import ctypes as C

import numpy as np

from . import device as D
from . import lib as _lib


def _clip(v, lo, hi):
    return np.minimum(np.maximum(v, lo), hi)


def epzs_search(org_plane, org_origin, s_org, ref_plane, ref_origin, s_ref, x, y, mvp, log2, bit_depth, lambda_mv, refi_bits, max_search_range,
                range_recentre, min_clip, max_clip, hpel_cnt, qpel_cnt, bi=0, org_bi=None, mv_start=None, extra_bits=0):
    """x, y: int arrays (block positions); mvp: int array [n, 2] (quarter pel, relative to the block).
    Returns (cost [n] uint32, mv [n, 2] int16), exactly what pinter_me_epzs returns per block."""
    n = len(x)
    x, y, mvp = np.asarray(x, np.int32), np.asarray(y, np.int32), np.asarray(mvp, np.int32).reshape(n, 2)
    S = 1 << log2
    P = _lib.MeParams(lambda_mv, refi_bits, extra_bits, bi, 3, max_search_range, range_recentre, (C.c_int32 * 2)(*min_clip),
                      (C.c_int32 * 2)(*max_clip), 0)
    sr = 5 if bi == 1 else range_recentre
    pos4 = np.stack([x << 2, y << 2], axis=1)
    gmvp = mvp + pos4

    def make_jobs(idx, start, tmpstep, clip_centre):
        j = np.zeros(len(idx), dtype=_lib.ME_JOB_DTYPE)
        j["x"], j["y"], j["org_off"] = x[idx], y[idx], idx * S * S
        cx, cy = x[idx] + (start[:, 0] >> 2), y[idx] + (start[:, 1] >> 2)
        if clip_centre:  # the first call clips the centre, the refinement calls do not (xeve_pinter.c:738-741 vs 785-788)
            cx, cy = _clip(cx, min_clip[0], max_clip[0]), _clip(cy, min_clip[1], max_clip[1])
        j["range"] = np.stack([_clip(cx - sr, min_clip[0], max_clip[0]), _clip(cy - sr, min_clip[1], max_clip[1]),
                               _clip(cx + sr, min_clip[0], max_clip[0]), _clip(cy + sr, min_clip[1], max_clip[1])], axis=1)
        j["gmvp"], j["mvi"], j["beststep_in"] = gmvp[idx], start + pos4[idx], tmpstep
        return j

    def near_mvp(mv, idx):
        return (np.abs(mvp[idx, 0] - mv[:, 0]) < 2) & (np.abs(mvp[idx, 1] - mv[:, 1]) < 2)

    all_idx = np.arange(n)
    start = np.asarray(mv_start, np.int32).reshape(n, 2) if bi == 1 else mvp
    res = D.me_ipel_diamond_jobs(org_plane, org_origin, s_org, org_bi, ref_plane, ref_origin, s_ref, make_jobs(all_idx, start, np.zeros(n, np.int32), True),
                                 log2, bit_depth, P)
    cost = res["cost"].copy()
    mv = res["mv"].astype(np.int32)
    tmpstep = res["beststep"].copy()
    beststep = np.where(near_mvp(mv, all_idx), 0, tmpstep)
    P.faststep = 2  # MAX_REFINE_SEARCH_STEP
    while bi != 1:
        act = np.nonzero(beststep > 0)[0]
        if len(act) == 0:
            break
        res = D.me_ipel_diamond_jobs(org_plane, org_origin, s_org, org_bi, ref_plane, ref_origin, s_ref, make_jobs(act, mv[act], tmpstep[act], False),
                                     log2, bit_depth, P)
        beststep[act] = 0
        tmpstep[act] = res["beststep"]
        better = res["cost"] < cost[act]
        bi_idx = act[better]
        cost[bi_idx] = res["cost"][better]
        mv[bi_idx] = res["mv"][better]
        beststep[bi_idx] = np.where(near_mvp(mv[bi_idx], bi_idx), 0, tmpstep[bi_idx])
    sj = np.zeros(n, dtype=_lib.SPEL_JOB_DTYPE)
    sj["x"], sj["y"], sj["org_off"], sj["gmvp"], sj["mvi"] = x, y, all_idx * S * S, gmvp, mv
    SP = _lib.SpelParams(lambda_mv, refi_bits, extra_bits, bi, hpel_cnt, qpel_cnt)
    res = D.me_spel_pattern_jobs(org_plane, org_origin, s_org, org_bi, ref_plane, ref_origin, s_ref, sj, log2, bit_depth, SP)
    better = res["cost"] < cost
    cost[better] = res["cost"][better]
    mv[better] = res["mv"][better]
    return cost, mv.astype(np.int16)


def epzs_search_device(org_plane, org_origin, s_org, ref_plane, ref_origin, s_ref, x, y, mvp, log2, bit_depth, lambda_mv, refi_bits,
                       max_search_range, range_recentre, min_clip, max_clip, hpel_cnt, qpel_cnt, bi=0, org_bi=None, mv_start=None, extra_bits=0,
                       with_mot_bits=False, raster=False, refi=0):
    """Same search through the C entry point xeve_hip_me_epzs_jobs: the bookkeeping between the searches runs in device
    kernels, the job / state / result arrays never leave the GPU until the final result.  raster: me_complexity > 1 (me_raster after a
    first search that ended far from its start; its step scales with refi + 1); hpel_cnt == 0: me_level = ME_LEV_IPEL (integer refinement
    instead of the sub-pel pattern)."""
    import torch

    L = _lib.load()
    n = len(x)
    S = 1 << log2
    jobs = np.zeros(n, dtype=_lib.EPZS_JOB_DTYPE)
    jobs["x"], jobs["y"], jobs["org_off"] = x, y, np.arange(n) * S * S
    jobs["mvp"] = np.asarray(mvp).reshape(n, 2)
    if mv_start is not None:
        jobs["mv_start"] = np.asarray(mv_start).reshape(n, 2)
    dev = org_plane.device
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(n, -1).copy()).to(dev)
    res = torch.empty((n, np.dtype(_lib.ME_RESULT_DTYPE).itemsize), dtype=torch.uint8, device=dev)
    ws_bytes = int(L.xeve_hip_me_epzs_workspace(n))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
    P = _lib.EpzsParams(_lib.MeParams(lambda_mv, refi_bits, extra_bits, bi, 3, max_search_range, range_recentre, (C.c_int32 * 2)(*min_clip),
                                      (C.c_int32 * 2)(*max_clip), (1 if raster else 0) | (int(refi) << 8)), hpel_cnt, qpel_cnt)
    coef = D.baseline_coef_l()
    _lib.check(L.xeve_hip_me_epzs_jobs(C.c_void_p(org_plane.data_ptr() + 2 * org_origin), s_org, C.c_void_p(org_bi.data_ptr()) if org_bi is not None else None,
                                       C.c_void_p(ref_plane.data_ptr() + 2 * ref_origin), s_ref, C.c_void_p(d_jobs.data_ptr()), n, log2, log2, bit_depth,
                                       C.c_void_p(coef.ctypes.data), C.byref(P), C.c_void_p(res.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    r = res.cpu().numpy().view(_lib.ME_RESULT_DTYPE).reshape(-1)
    if with_mot_bits:  # what the searches leave in pi->mot_bits[lidx]; 0 = untouched
        return r["cost"].copy(), r["mv"].copy(), r["best_mv_bits"].copy()
    return r["cost"].copy(), r["mv"].copy()
