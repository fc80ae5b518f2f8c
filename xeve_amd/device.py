"""Batched device API of libxeve_hip.so on torch-owned HBM buffers (torch = device memory + streams only).

Every function forwards device pointers, sizes and the CURRENT torch stream to the C-ABI
(include/xeve_hip.h, section 2) and returns torch tensors living on the GPU.  No arithmetic happens in
torch or on the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib

SBAC_BYTES = 36 + 2 * _lib.SBAC_NCTX  # sizeof(xeve_hip_sbac)

_COEF_L = None
_COEF_C = None


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i16(t):
    assert t.dtype == torch.int16 and t.is_cuda and t.is_contiguous()
    return t


def make_jobs(off1, off2, device):
    """xeve_hip_job[n] as an int32 [n, 2] device tensor."""
    j = np.stack([np.asarray(off1, np.int32), np.asarray(off2, np.int32)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(j)).to(device)


def make_mc_jobs(gmv_x, gmv_y, pred_off, frac, device):
    j = np.stack([np.asarray(a, np.int32) for a in (gmv_x, gmv_y, pred_off, frac)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(j)).to(device)


def _dist(fn, out_dtype, p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth, extra=()):
    L = _lib.load()
    njobs, ncand = jobs.shape[0], cand_off.numel()
    out = torch.empty((njobs, ncand), dtype=out_dtype, device=p1.device)
    _lib.check(getattr(L, fn)(_ptr(_i16(p1)), s1, _ptr(_i16(p2)), s2, _ptr(jobs), njobs, _ptr(cand_off), ncand, w, h, bit_depth,
                              *extra, _ptr(out), _stream()))
    return out


def sad_jobs(p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth, signed=False, out=None, mode=0):
    flags = (1 if signed else 0) | (mode << 4)
    if out is None:
        return _dist("xeve_hip_sad_jobs", torch.int32, p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth, (flags,))
    L = _lib.load()
    _lib.check(L.xeve_hip_sad_jobs(_ptr(p1), s1, _ptr(p2), s2, _ptr(jobs), jobs.shape[0], _ptr(cand_off), cand_off.numel(), w, h,
                                   bit_depth, flags, _ptr(out), _stream()))
    return out


def plane_shift1(plane):
    """copy of a plane shifted by one element (out.view(-1)[i] == plane.view(-1)[i + 1]); see xeve_hip_sad_jobs_dual"""
    out = torch.empty_like(plane)
    _lib.check(_lib.load().xeve_hip_plane_shift1(_ptr(_i16(plane)), _ptr(out), plane.numel(), _stream()))
    return out


def sad_jobs_dual(p1, s1, p2, p2_shift1, s2, jobs, cand_off, w, h, bit_depth, signed=False, out=None, tune=0):
    if out is None:
        out = torch.empty((jobs.shape[0], cand_off.numel()), dtype=torch.int32, device=p1.device)
    _lib.check(_lib.load().xeve_hip_sad_jobs_dual(_ptr(_i16(p1)), s1, _ptr(_i16(p2)), _ptr(_i16(p2_shift1)), s2, _ptr(jobs), jobs.shape[0],
                                                  _ptr(cand_off), cand_off.numel(), w, h, bit_depth, (1 if signed else 0) | (tune << 8), _ptr(out),
                                                  _stream()))
    return out


def ssd_jobs(p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth):
    return _dist("xeve_hip_ssd_jobs", torch.int64, p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth)


def satd_jobs(p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth):
    return _dist("xeve_hip_satd_jobs", torch.int32, p1, s1, p2, s2, jobs, cand_off, w, h, bit_depth)


def diff_jobs(p1, s1, p2, s2, jobs, w, h, out=None):
    L = _lib.load()
    if out is None:
        out = torch.empty((jobs.shape[0], h, w), dtype=torch.int16, device=p1.device)
    _lib.check(L.xeve_hip_diff_jobs(_ptr(_i16(p1)), s1, _ptr(_i16(p2)), s2, _ptr(jobs), jobs.shape[0], w, h, _ptr(out), _stream()))
    return out


# Baseline interpolation filters (reference: src_base/xeve_mc.c:39-93): quarter-pel rows of the 1/16-pel luma
# table and eighth-pel rows of the 1/32-pel chroma table; all other rows are zero.
def baseline_coef_l():
    global _COEF_L
    if _COEF_L is None:
        t = np.zeros((16, 8), np.int16)
        t[0], t[4], t[8], t[12] = [0, 0, 0, 64, 0, 0, 0, 0], [0, 1, -5, 52, 20, -5, 1, 0], [0, 2, -10, 40, 40, -10, 2, 0], [0, 1, -5, 20, 52, -5, 1, 0]
        _COEF_L = t
    return _COEF_L


def baseline_coef_c():
    global _COEF_C
    if _COEF_C is None:
        t = np.zeros((32, 4), np.int16)
        rows = [[0, 64, 0, 0], [-2, 58, 10, -2], [-4, 52, 20, -4], [-6, 46, 30, -6], [-8, 40, 40, -8], [-6, 30, 46, -6], [-4, 20, 52, -4], [-2, 10, 58, -2]]
        for i, r in enumerate(rows):
            t[4 * i] = r
        _COEF_C = t
    return _COEF_C


def mc_jobs(luma, ref, s_ref, pred, s_pred, jobs, w, h, bit_depth, coef=None):
    L = _lib.load()
    if coef is None:
        coef = baseline_coef_l() if luma else baseline_coef_c()
    fn = L.xeve_hip_mc_l_jobs if luma else L.xeve_hip_mc_c_jobs
    _lib.check(fn(_ptr(_i16(ref)), s_ref, _ptr(_i16(pred)), s_pred, _ptr(jobs), jobs.shape[0], w, h, bit_depth,
                  C.c_void_p(coef.ctypes.data), _stream()))
    return pred


def mc_l_sad_jobs(ref, s_ref, org, s_org, jobs, w, h, bit_depth, out, coef=None):
    """fused luma interpolation + SAD against the original (jobs[:, 2] = block offset inside `org`)"""
    coef = baseline_coef_l() if coef is None else coef
    _lib.check(_lib.load().xeve_hip_mc_l_sad_jobs(_ptr(_i16(ref)), s_ref, _ptr(_i16(org)), s_org, _ptr(jobs), jobs.shape[0], w, h, bit_depth,
                                                  C.c_void_p(coef.ctypes.data), _ptr(out), _stream()))
    return out


def mc_ssd_jobs(luma, ref, s_ref, org, s_org, jobs, w, h, bit_depth, out, coef=None):
    """fused interpolation + SSD against the original"""
    if coef is None:
        coef = baseline_coef_l() if luma else baseline_coef_c()
    _lib.check(_lib.load().xeve_hip_mc_ssd_jobs(int(luma), _ptr(_i16(ref)), s_ref, _ptr(_i16(org)), s_org, _ptr(jobs), jobs.shape[0], w, h, bit_depth,
                                                C.c_void_p(coef.ctypes.data), _ptr(out), _stream()))
    return out


def avg(a, b, out=None):
    L = _lib.load()
    if out is None:
        out = torch.empty_like(a)
    _lib.check(L.xeve_hip_avg(_ptr(_i16(a)), _ptr(_i16(b)), _ptr(out), a.numel(), _stream()))
    return out


def trans(coef, log2w, log2h, bit_depth):
    """in place on an int16 [nblk, h*w] tensor"""
    _lib.check(_lib.load().xeve_hip_trans(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, bit_depth, _stream()))
    return coef


def itrans(coef, log2w, log2h, bit_depth):
    _lib.check(_lib.load().xeve_hip_itrans(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, bit_depth, _stream()))
    return coef


def quant(coef, log2w, log2h, qp, scale, is_intra_slice, bit_depth, want_nnz=True):
    nnz = torch.empty(coef.shape[0], dtype=torch.int32, device=coef.device) if want_nnz else None
    _lib.check(_lib.load().xeve_hip_quant(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, qp, scale, int(is_intra_slice), bit_depth,
                                          _ptr(nnz) if want_nnz else None, _stream()))
    return nnz


def rdoq_zero_test(coef, log2w, log2h, qp, scale, is_intra_slice, bit_depth):
    coded = torch.empty(coef.shape[0], dtype=torch.int32, device=coef.device)
    _lib.check(_lib.load().xeve_hip_rdoq_zero_test(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, qp, scale, int(is_intra_slice),
                                                   bit_depth, _ptr(coded), _stream()))
    return coded


def rdoq(coef, log2w, log2h, qp, lam, is_luma, bit_depth, est, tool_iqt=0, nnz=None):
    """xeve_rdoq_run_length_cc over an int16 [nblk, h*w] tensor, in place (xeve_hip_rdoq); est: lib.RdoqEst"""
    if nnz is None:
        nnz = torch.empty(coef.shape[0], dtype=torch.int32, device=coef.device)
    _lib.check(_lib.load().xeve_hip_rdoq(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, qp, float(lam), int(is_luma), bit_depth, tool_iqt,
                                         C.byref(est), _ptr(nnz), _stream()))
    return nnz


def rdoq_bit_est(sbac):
    """xeve_rdoq_bit_est for an array of coder states (uint8 tensor of lib.SBAC_DTYPE records) -> int32 [n, 108] (xeve_hip_rdoq_est_full)"""
    n = sbac.numel() // SBAC_BYTES
    est = torch.empty((n, _lib.EST_FULL_INTS), dtype=torch.int32, device=sbac.device)
    _lib.check(_lib.load().xeve_hip_rdoq_bit_est(_ptr(sbac), n, _ptr(est), _stream()))
    return est


def rdoq_dev(coef, log2w, log2h, qp, lam, ch_type, bit_depth, est, est_idx=None, zero_test=False, is_intra_slice=False, is_intra_cu=False, tool_iqt=0,
             nnz=None):
    """RDOQ with per-block estimates from device memory (xeve_hip_rdoq_dev): est = rdoq_bit_est(...), est_idx int32 [nblk] or None"""
    if nnz is None:
        nnz = torch.empty(coef.shape[0], dtype=torch.int32, device=coef.device)
    _lib.check(_lib.load().xeve_hip_rdoq_dev(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, qp, float(lam), int(ch_type), bit_depth, tool_iqt,
                                             _ptr(est), _ptr(est_idx) if est_idx is not None else None, int(zero_test), int(is_intra_slice),
                                             int(is_intra_cu), _ptr(nnz), _stream()))
    return nnz


def mc_cu_jobs(refp, num_refp, s_l, s_c, pic_w, pic_h, jobs, w, h, bit_depth, chroma_format_idc=1, pred=None, workspace=None):
    """the CU motion-compensation driver xeve_mc for a batch of CUs (xeve_hip_mc_cu_jobs).  refp: HOST numpy array of
    lib.REFPIC_DTYPE records [refi * 2 + list] holding device addresses; jobs: uint8 tensor of lib.CU_MC_JOB_DTYPE records.
    Returns [pred_y, pred_u, pred_v] (dense per job)."""
    L = _lib.load()
    njobs = jobs.numel() // 20
    ws, hs = (1 if chroma_format_idc <= 2 else 0), (1 if chroma_format_idc <= 1 else 0)
    dev = jobs.device
    if pred is None:
        pred = [torch.empty((njobs, h * w), dtype=torch.int16, device=dev)] + [torch.empty((njobs, (h >> hs) * (w >> ws)), dtype=torch.int16, device=dev) for _ in range(2)]
    need = L.xeve_hip_mc_cu_workspace(njobs, w, h, num_refp[0], num_refp[1])
    if workspace is None:
        workspace = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
    cl, cc = C.c_void_p(baseline_coef_l().ctypes.data), C.c_void_p(baseline_coef_c().ctypes.data)
    _lib.check(L.xeve_hip_mc_cu_jobs(refp.ctypes.data_as(C.c_void_p), num_refp[0], num_refp[1], s_l, s_c, pic_w, pic_h, _ptr(jobs), njobs, w, h,
                                     bit_depth, bit_depth, chroma_format_idc, cl, cc, _ptr(pred[0]), _ptr(pred[1]), _ptr(pred[2]), _ptr(workspace),
                                     workspace.numel(), _stream()))
    return pred


def residue_rdo_jobs(org_ptrs, s_org_l, s_org_c, refp, s_l, s_c, states, params, jobs, workspace=None):
    """pinter_residue_rdo for a batch of candidates (xeve_hip_residue_rdo_jobs).  org_ptrs: three device addresses of sample (0, 0);
    refp: HOST numpy array of lib.REFPIC_DTYPE; states / jobs: uint8 tensors of lib.SBAC_DTYPE / lib.RDO_JOB_DTYPE records.
    Returns (results uint8 [njobs, 72], coef int16 flat, best uint8 [njobs, 180])."""
    L = _lib.load()
    njobs, nstates, dev = jobs.numel() // 36, states.numel() // SBAC_BYTES, jobs.device
    ws, hs = (1 if params.chroma_format_idc <= 2 else 0), (1 if params.chroma_format_idc <= 1 else 0)
    n0 = 1 << (params.log2_cuw + params.log2_cuh)
    n1 = (n0 >> (ws + hs)) if params.chroma_format_idc else 0
    res = torch.empty((njobs, 72), dtype=torch.uint8, device=dev)
    coef = torch.empty(max(1, njobs * (n0 + 2 * n1)), dtype=torch.int16, device=dev)
    best = torch.empty((njobs, SBAC_BYTES), dtype=torch.uint8, device=dev)
    need = L.xeve_hip_residue_rdo_workspace(njobs, nstates, C.byref(params), s_org_l, s_org_c)
    if workspace is None:
        workspace = torch.empty(int(need), dtype=torch.uint8, device=dev)
    org = (C.c_void_p * 3)(*[int(a) for a in org_ptrs])
    cl, cc = C.c_void_p(baseline_coef_l().ctypes.data), C.c_void_p(baseline_coef_c().ctypes.data)
    _lib.check(L.xeve_hip_residue_rdo_jobs(org, s_org_l, s_org_c, refp.ctypes.data_as(C.c_void_p), s_l, s_c, _ptr(states), nstates, C.byref(params), _ptr(jobs),
                                           njobs, cl, cc, _ptr(res), _ptr(coef), _ptr(best), _ptr(workspace), workspace.numel(), _stream()))
    return res, coef, best


def analyze_skip_jobs(org_ptrs, s_org_l, s_org_c, refp, s_l, s_c, states, params, jobs, max_cand=4, want_state=True, workspace=None):
    """xeve_analyze_skip for a batch of CUs (xeve_hip_analyze_skip_jobs).  jobs: uint8 tensor of lib.SKIP_JOB_DTYPE records.
    Returns (results uint8 [njobs, 40], pred_y, pred_u, pred_v int16 [njobs, n], best uint8 [njobs, 180] or None); predictions and
    states of CUs without a usable pair keep the zero fill."""
    L = _lib.load()
    njobs, nstates, dev = jobs.numel() // 60, states.numel() // SBAC_BYTES, jobs.device
    ws, hs = (1 if params.chroma_format_idc <= 2 else 0), (1 if params.chroma_format_idc <= 1 else 0)
    n0 = 1 << (params.log2_cuw + params.log2_cuh)
    n1 = (n0 >> (ws + hs)) if params.chroma_format_idc else 0
    res = torch.empty((njobs, 40), dtype=torch.uint8, device=dev)
    py = torch.zeros((njobs, n0), dtype=torch.int16, device=dev)
    pu, pv = (torch.zeros((njobs, max(n1, 1)), dtype=torch.int16, device=dev) for _ in range(2))
    best = torch.zeros((njobs, SBAC_BYTES), dtype=torch.uint8, device=dev) if want_state else None
    need = L.xeve_hip_analyze_skip_workspace(njobs, C.byref(params), max_cand)
    if workspace is None:
        workspace = torch.empty(int(need), dtype=torch.uint8, device=dev)
    org = (C.c_void_p * 3)(*[int(a) for a in org_ptrs])
    cl, cc = C.c_void_p(baseline_coef_l().ctypes.data), C.c_void_p(baseline_coef_c().ctypes.data)
    _lib.check(L.xeve_hip_analyze_skip_jobs(org, s_org_l, s_org_c, refp.ctypes.data_as(C.c_void_p), s_l, s_c, _ptr(states), nstates, C.byref(params), _ptr(jobs),
                                            njobs, max_cand, cl, cc, _ptr(res), _ptr(py), _ptr(pu), _ptr(pv), _ptr(best) if best is not None else None,
                                            _ptr(workspace), workspace.numel(), _stream()))
    return res, py, pu, pv, best


def pinter_analyze_cu_jobs(org_ptrs, s_org_l, s_org_c, refp, s_l, s_c, states, params, jobs, workspace=None, want_pred=False):
    """the whole inter analysis of a batch of CUs (xeve_hip_pinter_analyze_cu_jobs).  params: lib.InterParams; jobs: uint8 tensor of lib.INTER_JOB_DTYPE
    records.  Returns (results uint8 [njobs, 96], coef int16 flat [Y blocks | U blocks | V blocks], rec_y, rec_u, rec_v int16 [njobs, n],
    next_best uint8 [njobs, 180]) and, with want_pred, the winner's luma prediction int16 [njobs, n] (mi->pred_y_best)."""
    L = _lib.load()
    rp = params.rdo
    njobs, nstates, dev = jobs.numel() // 52, states.numel() // SBAC_BYTES, jobs.device
    ws, hs = (1 if rp.chroma_format_idc <= 2 else 0), (1 if rp.chroma_format_idc <= 1 else 0)
    n0 = 1 << (rp.log2_cuw + rp.log2_cuh)
    n1 = (n0 >> (ws + hs)) if rp.chroma_format_idc else 0
    res = torch.empty((njobs, 96), dtype=torch.uint8, device=dev)
    coef = torch.empty(max(1, njobs * (n0 + 2 * n1)), dtype=torch.int16, device=dev)
    ry = torch.zeros((njobs, n0), dtype=torch.int16, device=dev)
    ru, rv = (torch.zeros((njobs, max(n1, 1)), dtype=torch.int16, device=dev) for _ in range(2))
    nb = torch.zeros((njobs, SBAC_BYTES), dtype=torch.uint8, device=dev)
    py = torch.zeros((njobs, n0), dtype=torch.int16, device=dev) if want_pred else None
    need = L.xeve_hip_pinter_analyze_cu_workspace(njobs, nstates, C.byref(params), s_org_l, s_org_c)
    if workspace is None:
        workspace = torch.empty(int(need), dtype=torch.uint8, device=dev)
    org = (C.c_void_p * 3)(*[int(a) for a in org_ptrs])
    cl, cc = C.c_void_p(baseline_coef_l().ctypes.data), C.c_void_p(baseline_coef_c().ctypes.data)
    _lib.check(L.xeve_hip_pinter_analyze_cu_jobs(org, s_org_l, s_org_c, refp.ctypes.data_as(C.c_void_p), s_l, s_c, _ptr(states), nstates, C.byref(params),
                                                 _ptr(jobs), njobs, cl, cc, _ptr(res), _ptr(coef), _ptr(ry), _ptr(ru), _ptr(rv), _ptr(py) if py is not None else None, _ptr(nb),
                                                 _ptr(workspace), workspace.numel(), _stream()))
    return (res, coef, ry, ru, rv, nb, py) if want_pred else (res, coef, ry, ru, rv, nb)


def pintra_analyze_cu_jobs(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, states, params, jobs, pic_elems=None, workspace=None):
    """the intra analysis of a batch of CUs (xeve_hip_pintra_analyze_cu_jobs).  params: lib.IntraParams; jobs: uint8 tensor of lib.INTRA_JOB_DTYPE records; org_ptrs /
    mod_ptrs: device addresses of sample (0, 0) of the three planes of the original and of the picture being reconstructed; pic_elems: five element distances between
    the pictures of a multi-picture batch (org luma, org chroma, mod luma, mod chroma, maps) or None.  Returns (results uint8 [njobs, 32], coef int16 flat [Y | U | V
    blocks], rec int16 flat in the same layout, best uint8 [njobs, 180])."""
    L = _lib.load()
    njobs, nstates, dev = jobs.numel() // 24, states.numel() // SBAC_BYTES, jobs.device
    ws, hs = (1 if params.chroma_format_idc <= 2 else 0), (1 if params.chroma_format_idc <= 1 else 0)
    n0 = 1 << (params.log2_cuw + params.log2_cuh)
    n1 = (n0 >> (ws + hs)) if params.chroma_format_idc else 0
    res = torch.empty((njobs, 32), dtype=torch.uint8, device=dev)
    coef = torch.zeros(max(1, njobs * (n0 + 2 * n1)), dtype=torch.int16, device=dev)
    rec = torch.zeros(max(1, njobs * (n0 + 2 * n1)), dtype=torch.int16, device=dev)
    best = torch.zeros((njobs, SBAC_BYTES), dtype=torch.uint8, device=dev)
    need = L.xeve_hip_pintra_analyze_cu_workspace(njobs, nstates, C.byref(params))
    if need == 0 and njobs:
        raise _lib.XeveHipError("xeve_hip_pintra_analyze_cu_workspace: parameters outside the supported set")
    if workspace is None:
        workspace = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
    org = (C.c_void_p * 3)(*[int(a) for a in org_ptrs])
    mod = (C.c_void_p * 3)(*[int(a) for a in mod_ptrs])
    pe = (C.c_int64 * 5)(*[int(v) for v in pic_elems]) if pic_elems is not None else None
    _lib.check(L.xeve_hip_pintra_analyze_cu_jobs(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, _ptr(map_scu), _ptr(map_ipm), _ptr(map_tidx), pe, _ptr(states), nstates,
                                                 C.byref(params), _ptr(jobs), njobs, _ptr(res), _ptr(coef), _ptr(rec), _ptr(best), _ptr(workspace), workspace.numel(),
                                                 _stream()))
    return res, coef, rec, best


def mode_analyze_ctu_jobs(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, states, params, jobs, inter=None, pic_elems=None,
                          workspace=None, outputs=None):
    """the mode decision of a batch of CTUs (xeve_hip_mode_analyze_ctu_jobs): one chain per job, walked in lockstep.  params: lib.TreeParams; inter: lib.TreeInter for a
    P / B slice (its refp = address of a HOST table of lib.REFPIC_DTYPE records with device planes; map_mv / map_refi / col_mv*: device addresses; coef_l / coef_c are
    filled in here), None for an I slice; jobs: uint8 tensor of lib.CTU_JOB_DTYPE records.  The planes of the picture being reconstructed (mod_ptrs) and the maps are
    updated in place.  Returns (ctu data uint8 [nchains, 62976], next_best uint8 [nchains, 180], cost float64 [nchains])."""
    L = _lib.load()
    n, nstates, dev = jobs.numel() // 16, states.numel() // SBAC_BYTES, jobs.device
    if outputs is not None:  # the caller's own buffers (fixed addresses: what the library's graph replay is keyed on; no per-call zero fill)
        out, nxt, cost = outputs
        assert out.numel() >= n * _lib.CTU_DATA_BYTES and nxt.numel() >= n * SBAC_BYTES and cost.numel() >= n
    else:
        out = torch.zeros((n, _lib.CTU_DATA_BYTES), dtype=torch.uint8, device=dev)
        nxt = torch.zeros((n, SBAC_BYTES), dtype=torch.uint8, device=dev)
        cost = torch.zeros(n, dtype=torch.float64, device=dev)
    ip = None
    if inter is not None:
        inter.coef_l, inter.coef_c = baseline_coef_l().ctypes.data, baseline_coef_c().ctypes.data
        ip = C.byref(inter)
    need = L.xeve_hip_mode_analyze_ctu_workspace(n, C.byref(params), ip, s_org_l, s_org_c)
    if need == 0 and n:
        raise _lib.XeveHipError("xeve_hip_mode_analyze_ctu_workspace: parameters outside the supported set")
    if workspace is None:
        workspace = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
    org = (C.c_void_p * 3)(*[int(a) for a in org_ptrs])
    mod = (C.c_void_p * 3)(*[int(a) for a in mod_ptrs])
    pe = (C.c_int64 * 5)(*[int(v) for v in pic_elems]) if pic_elems is not None else None
    _lib.check(L.xeve_hip_mode_analyze_ctu_jobs(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, _ptr(map_scu), _ptr(map_ipm), _ptr(map_tidx), _ptr(map_cu_mode), pe,
                                                _ptr(states), nstates, C.byref(params), ip, _ptr(jobs), n, _ptr(out), _ptr(nxt), _ptr(cost), _ptr(workspace),
                                                workspace.numel(), _stream()))
    return out, nxt, cost


def eco_ctu_jobs(ctus, states, params, map_scu, map_ipm, map_tidx, map_cu_mode, jobs, map_pic_elems=0, bytes_cap=1 << 15, out=None):
    """the bitstream writer's side of a batch of decided CTUs (xeve_hip_eco_ctu_jobs): ctus = the ctu data tensor of mode_analyze_ctu_jobs; states (uint8 [nstates, 180])
    is advanced IN PLACE -- it is the writer's coder, and afterwards the entry state of each chain's next CTU; the maps receive the written CUs' flags.  params:
    lib.EcoParams.  Returns (bytes uint8 [nchains, bytes_cap], nbytes int32 [nchains])."""
    L = _lib.load()
    n, dev = jobs.numel() // 16, jobs.device
    if out is not None:
        by, nb = out
    else:
        by = torch.zeros((n, bytes_cap), dtype=torch.uint8, device=dev)
        nb = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(L.xeve_hip_eco_ctu_jobs(_ptr(ctus), _ptr(states), states.numel() // SBAC_BYTES, C.byref(params), _ptr(map_scu), _ptr(map_ipm), _ptr(map_tidx), _ptr(map_cu_mode),
                                       int(map_pic_elems), _ptr(jobs), n, _ptr(by), by.shape[1], _ptr(nb), _stream()))
    return by, nb


def eco_tile_end_jobs(states, jobs, bytes_cap=64):
    """the end of a tile on every chain's writer state (xeve_hip_eco_tile_end_jobs; states advanced in place).  Returns (bytes uint8 [nchains, bytes_cap], nbytes int32)."""
    n, dev = jobs.numel() // 16, jobs.device
    by = torch.zeros((n, bytes_cap), dtype=torch.uint8, device=dev)
    nb = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().xeve_hip_eco_tile_end_jobs(_ptr(states), states.numel() // SBAC_BYTES, _ptr(jobs), n, _ptr(by), bytes_cap, _ptr(nb), _stream()))
    return by, nb


def mode_analyze_ctu_intra_jobs(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, states, params, jobs, pic_elems=None,
                                workspace=None):
    """the I-slice form (chains may belong to different pictures: pic_elems)"""
    return mode_analyze_ctu_jobs(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, states, params, jobs, None, pic_elems, workspace)


def inter_candidates(map_scu, map_tidx, map_mv, col_mv0, col_mv1, w_scu, h_scu, log2_cuw, log2_cuh, slice_type, jobs):
    """fills jobs[].mvp / mv_col (uint8 tensor of lib.INTER_JOB_DTYPE records, in place) from the per-unit maps (xeve_hip_inter_candidates)"""
    _lib.check(_lib.load().xeve_hip_inter_candidates(_ptr(map_scu), _ptr(map_tidx) if map_tidx is not None else None, _ptr(map_mv), _ptr(col_mv0),
                                                     _ptr(col_mv1) if col_mv1 is not None else None, w_scu, h_scu, log2_cuw, log2_cuh, slice_type, _ptr(jobs),
                                                     jobs.numel() // 52, _stream()))
    return jobs


def _ptr_at(t, elem_off):
    return C.c_void_p(t.data_ptr() + int(elem_off) * t.element_size())


def deblock(planes, origins, s_l, s_c, map_scu, map_cu_mode, map_refi, map_mv, params, map_tidx=None):
    """in-loop deblocking of one picture in place (xeve_hip_deblock).  planes: three int16 tensors; origins: element offsets of
    sample (0, 0) in each; maps: device tensors laid out like the reference's per-4x4-unit arrays (map_tidx: uint8 tile index per unit, None = one
    tile); params: lib.DeblockParams"""
    _lib.check(_lib.load().xeve_hip_deblock(_ptr_at(_i16(planes[0]), origins[0]), _ptr_at(_i16(planes[1]), origins[1]), _ptr_at(_i16(planes[2]), origins[2]),
                                            s_l, s_c, _ptr(map_scu), _ptr(map_cu_mode), _ptr(map_tidx) if map_tidx is not None else None, _ptr(map_refi), _ptr(map_mv),
                                            C.byref(params), _stream()))


def picbuf_expand(planes, origins, s_l, s_c, w_l, h_l, w_c, h_c, exp_l, exp_c, chroma_format_idc=1):
    """xeve_picbuf_expand on planes resident in HBM (xeve_hip_picbuf_expand)"""
    _lib.check(_lib.load().xeve_hip_picbuf_expand(_ptr_at(_i16(planes[0]), origins[0]), _ptr_at(_i16(planes[1]), origins[1]), _ptr_at(_i16(planes[2]), origins[2]),
                                                  s_l, s_c, w_l, h_l, w_c, h_c, exp_l, exp_c, chroma_format_idc, _stream()))


def cu_bits_jobs(coef, sbac_in, jobs, params, want_state=True, workspace=None, bits=None, sbac_out=None):
    """CABAC bit count of inter-CU jobs (xeve_hip_cu_bits_jobs).  coef: flat int16 tensor; sbac_in / jobs: uint8 tensors holding
    arrays of lib.SBAC_DTYPE / lib.CU_BITS_JOB_DTYPE records; params: lib.CuBitsParams.  Returns (bits u32-as-int32 [njobs],
    exit states as a uint8 [njobs, 180] tensor or None)."""
    L = _lib.load()
    njobs = jobs.numel() // 44
    need = L.xeve_hip_cu_bits_workspace(njobs, coef.numel())
    if workspace is None:
        workspace = torch.empty(max(int(need), 4), dtype=torch.uint8, device=coef.device)
    if bits is None:
        bits = torch.empty(njobs, dtype=torch.int32, device=coef.device)
    if sbac_out is None and want_state:
        sbac_out = torch.empty((njobs, SBAC_BYTES), dtype=torch.uint8, device=coef.device)
    _lib.check(L.xeve_hip_cu_bits_jobs(_ptr(_i16(coef)), coef.numel(), _ptr(sbac_in), _ptr(jobs), njobs, C.byref(params), _ptr(workspace),
                                       workspace.numel(), _ptr(bits), _ptr(sbac_out) if sbac_out is not None else None, _stream()))
    return bits, sbac_out


def dquant(coef, log2w, log2h, scale, bit_depth):
    _lib.check(_lib.load().xeve_hip_dquant(_ptr(_i16(coef)), coef.shape[0], log2w, log2h, scale, bit_depth, _stream()))
    return coef


def recon(coef, pred, is_coef, cuw, cuh, rec_off, s_rec, rec, bit_depth):
    _lib.check(_lib.load().xeve_hip_recon(_ptr(_i16(coef)), _ptr(_i16(pred)), _ptr(is_coef) if is_coef is not None else None,
                                          coef.shape[0], cuw, cuh, _ptr(rec_off), s_rec, _ptr(_i16(rec)), bit_depth, _stream()))
    return rec


def residual_rdo(org, s_org, pred, s_pred, jobs, log2w, log2h, bit_depth, qp, is_intra_slice, zero_test, coef, rec, s_rec, nnz, ssd):
    """fused DIFF/SSD/DCT/quant/dequant/IDCT/recon/SSD (xeve_hip_residual_rdo); all outputs preallocated by the caller"""
    _lib.check(_lib.load().xeve_hip_residual_rdo(_ptr(_i16(org)), s_org, _ptr(_i16(pred)), s_pred, _ptr(jobs), jobs.shape[0], log2w, log2h,
                                                 bit_depth, qp, QUANT_SCALE[0][qp % 6], DQ_SCALE[qp % 6] << (qp // 6), int(is_intra_slice),
                                                 int(zero_test), _ptr(_i16(coef)), _ptr(_i16(rec)), s_rec, _ptr(nnz), _ptr(ssd), _stream()))


def me_ipel_diamond_jobs(org_plane, org_origin, s_org, org_bi, ref_plane, ref_origin, s_ref, jobs_np, log2, bit_depth, params):
    """One complete me_ipel_diamond per job (xeve_hip_me_ipel_diamond_jobs).  org_plane / ref_plane are padded int16 planes,
    *_origin the element offset of picture sample (0,0) inside them; jobs_np a numpy array of lib.ME_JOB_DTYPE; params a
    lib.MeParams.  Returns a numpy array of lib.ME_RESULT_DTYPE."""
    L = _lib.load()
    dev = org_plane.device
    jobs = torch.from_numpy(jobs_np.view(np.uint8).reshape(len(jobs_np), -1).copy()).to(dev)
    res = torch.empty((len(jobs_np), np.dtype(_lib.ME_RESULT_DTYPE).itemsize), dtype=torch.uint8, device=dev)
    _lib.check(L.xeve_hip_me_ipel_diamond_jobs(C.c_void_p(_i16(org_plane).data_ptr() + 2 * org_origin), s_org,
                                               _ptr(_i16(org_bi)) if org_bi is not None else None,
                                               C.c_void_p(_i16(ref_plane).data_ptr() + 2 * ref_origin), s_ref, _ptr(jobs), len(jobs_np), log2, log2,
                                               bit_depth, C.byref(params), _ptr(res), _stream()))
    return res.cpu().numpy().view(_lib.ME_RESULT_DTYPE).reshape(-1)


def me_spel_pattern_jobs(org_plane, org_origin, s_org, org_bi, ref_plane, ref_origin, s_ref, jobs_np, log2, bit_depth, params, coef=None):
    """One complete me_spel_pattern per job (xeve_hip_me_spel_pattern_jobs); jobs_np: numpy array of lib.SPEL_JOB_DTYPE."""
    L = _lib.load()
    dev = org_plane.device
    coef = baseline_coef_l() if coef is None else coef
    jobs = torch.from_numpy(jobs_np.view(np.uint8).reshape(len(jobs_np), -1).copy()).to(dev)
    res = torch.empty((len(jobs_np), np.dtype(_lib.ME_RESULT_DTYPE).itemsize), dtype=torch.uint8, device=dev)
    ws_bytes = int(L.xeve_hip_me_spel_workspace(len(jobs_np)))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.check(L.xeve_hip_me_spel_pattern_jobs(C.c_void_p(_i16(org_plane).data_ptr() + 2 * org_origin), s_org,
                                               _ptr(_i16(org_bi)) if org_bi is not None else None,
                                               C.c_void_p(_i16(ref_plane).data_ptr() + 2 * ref_origin), s_ref, _ptr(jobs), len(jobs_np), log2, log2,
                                               bit_depth, C.c_void_p(coef.ctypes.data), C.byref(params), _ptr(res), _ptr(ws), ws_bytes, _stream()))
    return res.cpu().numpy().view(_lib.ME_RESULT_DTYPE).reshape(-1)


def residual_rdoq(org, s_org, pred, s_pred, jobs, log2w, log2h, bit_depth, qp, is_intra_slice, lam, is_luma, est, coef, rec, s_rec, nnz, ssd,
                  tool_iqt=0):
    """the residual chain with zero pre-test + RDOQ as quantiser (xeve_hip_residual_rdoq)"""
    _lib.check(_lib.load().xeve_hip_residual_rdoq(_ptr(_i16(org)), s_org, _ptr(_i16(pred)), s_pred, _ptr(jobs), jobs.shape[0], log2w, log2h,
                                                  bit_depth, qp, QUANT_SCALE[tool_iqt][qp % 6], DQ_SCALE[qp % 6] << (qp // 6), int(is_intra_slice),
                                                  float(lam), int(is_luma), tool_iqt, C.byref(est), _ptr(_i16(coef)), _ptr(_i16(rec)), s_rec,
                                                  _ptr(nnz), _ptr(ssd), _stream()))


# quantiser scale tables of the standard (reference: src_base/xeve_tq.c:37-38, xeve_tbl.c:237)
QUANT_SCALE = ((26214, 23302, 20560, 18396, 16384, 14764), (26214, 23302, 20560, 18396, 16384, 14564))
DQ_SCALE = (40, 45, 51, 57, 64, 71)
