"""The whole inter analysis of a CU on the GPU (xeve_hip_pinter_analyze_cu_jobs = ctx->fn_pinter_analyze_cu for a batch): cost (bit pattern of the
double), cu_mode, motion data, core->nnz, coefficients, reconstruction and core->s_next_best against the reference goldens and the pinned oracle,
through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from _inter_cases import make_inter_jobs, make_inter_params, make_inter_picture, mask_unobservable
from _inter_golden import CASES, golden
from _libs import INTER_RESULT_DTYPE, SBAC_DTYPE, oracle_inter, ptr
from _mc_cases import refpic_table
from _rdo_cases import states

pytestmark = pytest.mark.gpu


def hip_params(P):
    """tests' InterParams (the oracle's layout) -> lib.InterParams (the library's: xeve_hip_epzs_params holds the sub-pel counts directly)"""
    from xeve_amd import lib

    H = lib.InterParams()
    C.memmove(C.byref(H.rdo), C.byref(P.rdo), C.sizeof(P.rdo))
    C.memmove(C.byref(H.me.me), C.byref(P.me), C.sizeof(P.me))
    H.me.hpel_cnt, H.me.qpel_cnt = P.spel.hpel_cnt, P.spel.qpel_cnt
    for l in range(2):
        for i in range(8):
            H.refi_bits[l][i], H.range_recentre[l][i] = P.refi_bits[l][i], P.range_recentre[l][i]
    H.max_cand, H.poc, H.col_list_poc0, H.skip_th = P.max_cand, P.poc, P.col_list_poc0, P.skip_th
    return H


def run_hip(refs, org, st, P, jobs):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
    lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
    dorg = [torch.from_numpy(x).to(dev) for x in org]
    org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
    res, coef, ry, ru, rv, nb = D.pinter_analyze_cu_jobs(org_ptrs, refs["s_l"], refs["s_c"], dev_tab, refs["s_l"], refs["s_c"],
                                                         torch.from_numpy(st.view(np.uint8).copy()).to(dev), hip_params(P),
                                                         torch.from_numpy(jobs.view(np.uint8).copy()).to(dev))
    torch.cuda.synchronize()
    n = len(jobs)
    n0 = 1 << (2 * P.rdo.log2_cuw)
    n1 = (n0 >> (refs["ws"] + refs["hs"])) if P.rdo.chroma_format_idc else 0
    c = coef.cpu().numpy()
    return (res.cpu().numpy().reshape(-1).view(INTER_RESULT_DTYPE), [c[:n * n0].reshape(n, n0), c[n * n0:n * (n0 + n1)].reshape(n, n1), c[n * (n0 + n1):n * (n0 + 2 * n1)].reshape(n, n1)],
            [ry.cpu().numpy(), ru.cpu().numpy(), rv.cpu().numpy()], nb.cpu().numpy().reshape(-1).view(SBAC_DTYPE))


def test_hip_pinter_analyze_cu_matches_reference_goldens():
    n = 0
    for c in golden():
        res, coef, rec, best = run_hip(c["refs"], c["org"], c["states"], c["P"], c["jobs"])
        got = mask_unobservable(res, c["slice_type"])
        for i in range(len(res)):
            assert got[i:i + 1].tobytes() == c["res"][i:i + 1].tobytes(), (n, i, res[i], c["res"][i])
        for k in range(3 if c["idc"] else 1):
            assert np.array_equal(coef[k], c["coef"][k]), (n, k)
            assert np.array_equal(rec[k], c["rec"][k]), (n, k)
        assert best.tobytes() == c["best"].tobytes(), n
        n += 1
    assert n == len(CASES)


@pytest.mark.parametrize("w,h,bd,nref,idc,slice_type,skip_th", [(128, 96, 10, 2, 1, 0, 0.0), (128, 64, 10, 2, 1, 1, 0.0), (96, 64, 8, 1, 1, 0, 0.0),
                                                                (64, 64, 10, 2, 0, 0, 0.0), (192, 128, 10, 3, 1, 0, 0.0), (128, 96, 10, 2, 1, 0, 6.0), (96, 64, 10, 2, 3, 0, 0.0),
                                                                (128, 128, 10, 4, 1, 1, 0.0)])
def test_hip_pinter_analyze_cu_vs_oracle(w, h, bd, nref, idc, slice_type, skip_th):
    O = oracle_inter()
    r = np.random.default_rng(13 * w + h + bd + nref + idc + slice_type)
    refs, org = make_inter_picture(r, w, h, bd, nref, idc, slice_type)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 7)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                        np.uint64)
    modes = set()
    for lw in (3, 4, 5, 6):
        cu = 1 << lw
        P = make_inter_params(r, lw, w, h, bd, nref, idc, slice_type, refs, skip_th)
        jobs = make_inter_jobs(r, 70, w, h, cu, len(st), refs, slice_type)
        res, coef, rec, best = run_hip(refs, org, st, P, jobs)
        nc = max(1, (cu >> refs["ws"]) * (cu >> refs["hs"]))
        for i in range(len(jobs)):
            er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ec = [np.zeros(cu * cu, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            ep = [x.copy() for x in ec]
            O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]),
                                   ptr(ec[1]), ptr(ec[2]), ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
            key = (lw, i, jobs[i], res[i], er[0])
            assert res[i:i + 1].tobytes() == er.tobytes(), key
            for k in range(3 if idc else 1):
                assert np.array_equal(coef[k][i], ec[k]) and np.array_equal(rec[k][i], ep[k]), (k,) + key
            assert best[i:i + 1].tobytes() == eb.tobytes(), key
            modes.add(int(er["best_idx"][0]))
    assert len(modes) >= (4 if slice_type == 0 else 2), modes


def test_hip_pinter_analyze_cu_edge_cases():
    """empty batch; parameters outside the path are refused with an error code and a message (no silent fallback); one CU at the picture corner"""
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    r = np.random.default_rng(4)
    refs, org = make_inter_picture(r, 64, 64, 10, 1, 1, 0)
    st = states(r, 2)
    P = make_inter_params(r, 6, 64, 64, 10, 1, 1, 0, refs)
    jobs = make_inter_jobs(r, 1, 64, 64, 64, len(st), refs, 0)
    res, coef, rec, best = run_hip(refs, org, st, P, jobs)  # the whole picture is one 64x64 CU
    O = oracle_inter()
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
    ec = [np.zeros(4096, np.int16), np.zeros(1024, np.int16), np.zeros(1024, np.int16)]
    ep = [x.copy() for x in ec]
    O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs), ptr(er), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]),
                           ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
    assert res.tobytes() == er.tobytes() and np.array_equal(rec[0][0], ep[0]) and best.tobytes() == eb.tobytes()
    # empty batch
    H = hip_params(P)
    dst = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
    empty = torch.empty(0, dtype=torch.uint8, device=dev)
    out = D.pinter_analyze_cu_jobs([1, 1, 1], refs["s_l"], refs["s_c"], np.zeros(2, lib.REFPIC_DTYPE), refs["s_l"], refs["s_c"], dst, H, empty)
    assert out[0].numel() == 0
    # refused: 4x4 CUs, more list-1 than list-0 pictures
    for mutate in (lambda h: setattr(h.rdo, "log2_cuw", 2), lambda h: h.rdo.num_refp.__setitem__(1, 3)):
        H = hip_params(P)
        mutate(H)
        with pytest.raises(xeve_amd.XeveHipError):
            D.pinter_analyze_cu_jobs([1, 1, 1], refs["s_l"], refs["s_c"], np.zeros(8, lib.REFPIC_DTYPE), refs["s_l"], refs["s_c"], dst, H,
                                     torch.zeros(52, dtype=torch.uint8, device=dev), workspace=torch.empty(1 << 20, dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("slice_type,tiles", [(0, 1), (1, 1), (0, 2)])
def test_hip_inter_candidates_vs_oracle(slice_type, tiles):
    """xeve_hip_inter_candidates: every CU position of a 192x128 picture at every size, against the pinned oracle"""
    import torch

    import xeve_amd
    from _inter_cases import make_maps
    from _libs import INTER_JOB_DTYPE, oracle_cand
    from xeve_amd import device as D

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O = oracle_cand()
    r = np.random.default_rng(41 + slice_type + tiles)
    w_scu, h_scu = 48, 32
    map_scu, tidx, map_mv, c0, c1 = make_maps(r, w_scu, h_scu, tiles)
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev)
    d = [up(map_scu), up(tidx) if tiles > 1 else None, up(map_mv), up(c0), up(c1)]
    for lw in (3, 4, 5, 6):
        s = 1 << (lw - 2)
        xs, ys = np.meshgrid(np.arange(w_scu // s) * s * 4, np.arange(h_scu // s) * s * 4)
        jobs = np.zeros(xs.size, INTER_JOB_DTYPE)
        jobs["x"], jobs["y"], jobs["sbac"], jobs["ctx_skip"] = xs.ravel(), ys.ravel(), 7, 1
        dj = up(jobs)
        D.inter_candidates(d[0], d[1], d[2], d[3], d[4] if slice_type == 0 else None, w_scu, h_scu, lw, lw, slice_type, dj)
        got = dj.cpu().numpy().view(INTER_JOB_DTYPE)
        for i in range(len(jobs)):
            e = jobs[i:i + 1].copy()
            O.xo_inter_candidates(ptr(map_scu), ptr(tidx), ptr(map_mv), ptr(c0), ptr(c1), w_scu, h_scu, lw, lw, slice_type, ptr(e))
            assert got[i:i + 1].tobytes() == e.tobytes(), (lw, i, got[i], e[0])


def test_hip_pinter_analyze_cu_full_size_properties():
    """every 16x16 CU of a 1920x1088 B picture (8160 CUs): the results do not depend on how the batch is cut or on the order the bi rounds hand out
    their slots (two runs, whole vs two halves, are identical); the returned cost is the minimum of the evaluated modes and belongs to the winner;
    a skipped CU has no coefficients; a sample of CUs agrees with the oracle"""
    w, h, bd, nref, lw = 1920, 1088, 10, 2, 4
    r = np.random.default_rng(77)
    refs, org = make_inter_picture(r, w, h, bd, nref, 1, 0)
    st = states(r, 16)
    P = make_inter_params(r, lw, w, h, bd, nref, 1, 0, refs, 0.0, max_cand=3)
    n = (w // 16) * (h // 16)
    jobs = make_inter_jobs(r, n, w, h, 16, len(st), refs, 0)
    jobs["x"], jobs["y"] = (np.arange(n) % (w // 16)) * 16, (np.arange(n) // (w // 16)) * 16
    a = run_hip(refs, org, st, P, jobs)
    b = run_hip(refs, org, st, P, jobs)
    c0, c1 = run_hip(refs, org, st, P, jobs[:n // 2]), run_hip(refs, org, st, P, jobs[n // 2:])
    assert a[0].tobytes() == b[0].tobytes() and a[3].tobytes() == b[3].tobytes()
    assert a[0].tobytes() == c0[0].tobytes() + c1[0].tobytes() and a[3].tobytes() == c0[3].tobytes() + c1[3].tobytes()
    for k in range(3):
        assert np.array_equal(a[1][k], b[1][k]) and np.array_equal(a[2][k], b[2][k])
        assert np.array_equal(a[1][k], np.concatenate([c0[1][k], c1[1][k]])) and np.array_equal(a[2][k], np.concatenate([c0[2][k], c1[2][k]]))
    res = a[0]
    assert (res["cost"] == res["cost_inter"].min(axis=1)).all() and (res["cost"] == res["cost_inter"][np.arange(n), res["best_idx"]]).all()
    assert (res["cost_inter"] < 1e300).all()  # skip_th 0 and a noisy picture: every mode was evaluated everywhere
    skip = res["cu_mode"] == 2
    assert skip.any() and (~skip).any() and not res["nnz"][skip].any() and not a[1][0][skip].any()
    assert set(np.unique(res["best_idx"])) >= {0, 1, 2, 3}
    O = oracle_inter()
    tab = refpic_table(refs, lambda x, off: int(x.ctypes.data) + 2 * off)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    for i in r.integers(0, n, size=120):
        er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
        ec = [np.zeros(256, np.int16), np.zeros(64, np.int16), np.zeros(64, np.int16)]
        ep = [x.copy() for x in ec]
        O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]),
                               ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
        assert res[i:i + 1].tobytes() == er.tobytes() and np.array_equal(a[2][0][i], ep[0]) and a[3][i:i + 1].tobytes() == eb.tobytes(), i


def test_hip_pinter_analyze_cu_list1_shorter_than_list0():
    O = oracle_inter()
    w, h, bd, nref = 128, 96, 10, 3
    r = np.random.default_rng(780)
    refs, org = make_inter_picture(r, w, h, bd, nref, 1, 0)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 5)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]], np.uint64)
    modes = set()
    for lw in (3, 4, 5):
        cu = 1 << lw
        P = make_inter_params(r, lw, w, h, bd, nref, 1, 0, refs, 0.0, nref1=2)
        jobs = make_inter_jobs(r, 60, w, h, cu, len(st), refs, 0)
        res, coef, rec, best = run_hip(refs, org, st, P, jobs)
        nc = (cu // 2) ** 2
        for i in range(len(jobs)):
            er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ec = [np.zeros(cu * cu, np.int16), np.zeros(nc, np.int16), np.zeros(nc, np.int16)]
            ep = [x.copy() for x in ec]
            O.xo_pinter_analyze_cu(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), P, ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]), ptr(ec[1]),
                                   ptr(ec[2]), ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
            assert res[i:i + 1].tobytes() == er.tobytes() and np.array_equal(rec[0][i], ep[0]) and best[i:i + 1].tobytes() == eb.tobytes(), (lw, i, res[i], er[0])
            modes.add(int(er["best_idx"][0]))
    assert len(modes) >= 3, modes


def test_hip_pinter_analyze_cu_fuzz():
    """tools/fuzz_inter.py, a short run: random configurations incl. 12-bit, 4:4:4, QP / lambda extremes, far candidates on picture corners"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_inter.py"), "8", "900"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "0 mismatches" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
