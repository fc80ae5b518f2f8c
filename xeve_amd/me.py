"""Per-list motion search of one picture level on the GPU: numpy-facing wrapper of xeve_hip_me_epzs_jobs -- pinter_me_epzs (reference:
src_base/xeve_pinter.c:699-869) for every block of a quad-tree level, the loop between the integer and sub-pel searches on the device too.
(The round-1 host-side composition of the same loop over the two search kernels is gone: the device entry point superseded it.)
"""
import ctypes as C

import numpy as np

from . import device as D
from . import lib as _lib


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
