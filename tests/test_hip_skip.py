"""xeve_analyze_skip on the GPU (xeve_hip_analyze_skip_jobs): cost (bit pattern of the double), winning pair, vectors, best SSD, the kept
prediction and core->s_temp_best against the reference goldens and the pinned oracle, through the C-ABI."""
import numpy as np
import pytest

from _libs import SBAC_DTYPE, SKIP_RESULT_DTYPE, oracle_skip, ptr
from _mc_cases import refpic_table
from _rdo_cases import make_params, make_picture, make_skip_jobs, states
from _skip_golden import N_CASES, golden

pytestmark = pytest.mark.gpu


def run_hip(refs, org, st, p, jobs, max_cand=4, want_state=True):
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
    hp = lib.RdoParams.from_buffer_copy(bytes(p))
    res, py, pu, pv, best = D.analyze_skip_jobs(org_ptrs, refs["s_l"], refs["s_c"], dev_tab, refs["s_l"], refs["s_c"],
                                                torch.from_numpy(st.view(np.uint8).copy()).to(dev), hp, torch.from_numpy(jobs.view(np.uint8).copy()).to(dev),
                                                max_cand=max_cand, want_state=want_state)
    torch.cuda.synchronize()
    return (res.cpu().numpy().reshape(-1).view(SKIP_RESULT_DTYPE), [py.cpu().numpy(), pu.cpu().numpy(), pv.cpu().numpy()],
            best.cpu().numpy().reshape(-1).view(SBAC_DTYPE) if best is not None else None)


def test_hip_analyze_skip_matches_reference_goldens():
    n = 0
    for c in golden():
        for max_cand in sorted({c["ncand"], 4}):
            res, pred, best = run_hip(c["refs"], c["org"], c["states"], c["p"], c["jobs"], max_cand)
            if c["slice_type"] != 0:
                res["mv"][:, 1] = 0
            assert res.tobytes() == c["res"].tobytes(), (n, np.flatnonzero(res["cost"] != c["res"]["cost"])[:5])
            for k in range(3 if c["idc"] else 1):
                assert np.array_equal(pred[k], c["pred"][k]), (n, k)
            assert best.tobytes() == c["best"].tobytes(), n
        n += 1
    assert n == N_CASES


@pytest.mark.parametrize("w,h,bd,nref,idc,slice_type,ncand", [(128, 96, 10, 2, 1, 0, 3), (128, 64, 10, 1, 1, 1, 2), (96, 64, 8, 3, 1, 0, 4), (64, 64, 10, 1, 0, 0, 3),
                                                              (256, 128, 10, 3, 1, 1, 4)])
def test_hip_analyze_skip_vs_oracle(w, h, bd, nref, idc, slice_type, ncand):
    """beyond what the reference's Baseline candidate derivation produces: any reference index per candidate, unusable ones (-1), CUs without a
    usable pair, per-CU candidate counts below the batch's"""
    O = oracle_skip()
    r = np.random.default_rng(7 * w + h + bd + nref + idc + slice_type + ncand)
    refs, org = make_picture(r, w, h, bd, nref, idc)
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    st = states(r, 7)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                        np.uint64)
    pairs, none = set(), 0
    for (lw, lh) in [(3, 3), (4, 4), (5, 5), (6, 6), (2, 2), (4, 3), (3, 5)]:
        cuw, cuh = 1 << lw, 1 << lh
        if cuw > w or cuh > h:
            continue
        p = make_params(r, lw, lh, w, h, bd, nref, idc, slice_type)
        jobs = make_skip_jobs(r, 80, w, h, cuw, cuh, len(st), ncand)
        jobs["refi_pred"] = r.integers(-1, nref, size=jobs["refi_pred"].shape)
        jobs["refi_pred"][:6] = -1
        jobs["ncand"][6:] = r.integers(1, ncand + 1, size=len(jobs) - 6)
        res, pred, best = run_hip(refs, org, st, p, jobs, ncand)
        nc = (cuw >> refs["ws"]) * (cuh >> refs["hs"])
        for i in range(len(jobs)):
            er, eb = np.zeros(1, SKIP_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ep = [np.zeros(cuw * cuh, np.int16), np.zeros(max(nc, 1), np.int16), np.zeros(max(nc, 1), np.int16)]
            O.xo_analyze_skip(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[i:i + 1]), ptr(er), ptr(ep[0]),
                              ptr(ep[1]), ptr(ep[2]), ptr(eb))
            key = (lw, lh, i, jobs[i], res[i], er[0])
            assert res[i:i + 1].tobytes() == er.tobytes(), key
            for k in range(3 if idc else 1):
                assert np.array_equal(pred[k][i], ep[k]), (k,) + key
            assert best[i:i + 1].tobytes() == eb.tobytes(), key
            if er["cost"][0] > 1e300:
                none += 1
            else:
                pairs.add((int(er["idx0"][0]), int(er["idx1"][0])))
    assert none >= 6 and len(pairs) >= (4 if slice_type == 0 else 2), (none, pairs)


def test_hip_analyze_skip_without_state_and_properties():
    """1080p-sized batch of 16x16 CUs: results do not depend on whether the coder state is requested; the winner's cost is reproduced from its
    own SSD and bits; duplicated candidates never win"""
    w, h, bd, nref, idc = 1920, 1088, 10, 2, 1
    r = np.random.default_rng(99)
    refs, org = make_picture(r, w, h, bd, nref, idc)
    st = states(r, 9)
    p = make_params(r, 4, 4, w, h, bd, nref, idc, 0)
    n = (w // 16) * (h // 16)
    jobs = make_skip_jobs(r, n, w, h, 16, 16, len(st), 4)
    jobs["x"], jobs["y"] = (np.arange(n) % (w // 16)) * 16, (np.arange(n) // (w // 16)) * 16
    a, pa, ba = run_hip(refs, org, st, p, jobs, 4, True)
    b, pb, bb = run_hip(refs, org, st, p, jobs, 4, False)
    assert a.tobytes() == b.tobytes() and bb is None and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    assert (a["cost"] < 1e300).all()
    i0, i1 = a["idx0"], a["idx1"]
    mv = jobs["mvp"]
    for l, idx in ((0, i0), (1, i1)):
        for k in range(3):
            earlier = (k < idx) & (mv[np.arange(n), l, k] == mv[np.arange(n), l, idx]).all(axis=1)
            assert not earlier.any()
    assert np.array_equal(a["mv"][:, 0], mv[np.arange(n), 0, i0]) and np.array_equal(a["mv"][:, 1], mv[np.arange(n), 1, i1])
    O = oracle_skip()
    tab = refpic_table(refs, lambda a_, off: int(a_.ctypes.data) + 2 * off)
    org_ptrs = np.array([int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]],
                        np.uint64)
    for t in r.integers(0, n, size=150):
        er, eb = np.zeros(1, SKIP_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
        ep = [np.zeros(256, np.int16), np.zeros(64, np.int16), np.zeros(64, np.int16)]
        O.xo_analyze_skip(ptr(org_ptrs), refs["s_l"], refs["s_c"], ptr(tab), refs["s_l"], refs["s_c"], ptr(st), p, ptr(jobs[t:t + 1]), ptr(er), ptr(ep[0]),
                          ptr(ep[1]), ptr(ep[2]), ptr(eb))
        assert a[t:t + 1].tobytes() == er.tobytes() and ba[t:t + 1].tobytes() == eb.tobytes(), (t, a[t], er[0])
        assert all(np.array_equal(pa[k][t], ep[k]) for k in range(3)), t
