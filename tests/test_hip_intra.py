"""pintra_analyze_cu on the GPU (xeve_hip_pintra_analyze_cu_jobs): cost (bit pattern of the double), distortion, prediction modes, core->nnz, coefficients,
reconstruction and core->s_temp_best against the reference goldens and the pinned oracle, through the C-ABI; the intra CU syntax of the bit counter."""
import ctypes as C

import numpy as np
import pytest

from _intra_cases import CASES, INTRA_RESULT_DTYPE, N_JOBS, golden, make_case, run_oracle, same
from _libs import SBAC_DTYPE

pytestmark = pytest.mark.gpu


def run_hip(c, pics=None):
    """all N_JOBS of a case in ONE call: every job has its own maps, so the batch is a multi-picture one (picture i = job i's maps, the planes shared = distance 0)"""
    import torch
    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    org = [torch.from_numpy(p.copy()).to(dev) for p in c["org"]]
    mod = [torch.from_numpy(p.copy()).to(dev) for p in c["mod"]]
    n_map = c["maps"][0][0].size
    ms = torch.from_numpy(np.concatenate([m[0] for m in c["maps"]]).view(np.int32)).to(dev)
    mi = torch.from_numpy(np.concatenate([m[1] for m in c["maps"]])).to(dev)
    mt = torch.from_numpy(np.concatenate([m[2] for m in c["maps"]])).to(dev)
    jobs = c["jobs"].copy()
    jobs["pic"] = np.arange(len(jobs))
    P = lib.IntraParams.from_buffer_copy(bytes(c["P"]))
    res, coef, rec, best = D.pintra_analyze_cu_jobs([t.data_ptr() for t in org], c["org"][0].shape[1], c["org"][1].shape[1], [t.data_ptr() for t in mod],
                                                    c["mod"][0].shape[1], c["mod"][1].shape[1], ms, mi, mt, torch.from_numpy(c["states"].view(np.uint8).copy()).to(dev), P,
                                                    torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), pic_elems=(0, 0, 0, 0, n_map))
    torch.cuda.synchronize()
    n, n0, n1 = len(jobs), c["n0"], c["n1"]
    res = res.cpu().numpy().reshape(-1).view(INTRA_RESULT_DTYPE)
    coef, rec, best = coef.cpu().numpy(), rec.cpu().numpy(), best.cpu().numpy().reshape(-1).view(SBAC_DTYPE)
    out = []
    for i in range(n):
        blocks = lambda a: [a[i * n0:(i + 1) * n0], a[n * n0 + i * n1:n * n0 + (i + 1) * n1] if n1 else np.zeros(1, np.int16),
                            a[n * (n0 + n1) + i * n1:n * (n0 + n1) + (i + 1) * n1] if n1 else np.zeros(1, np.int16)]
        out.append((res[i:i + 1], blocks(coef), blocks(rec), best[i:i + 1]))
    return out


def test_hip_intra_matches_reference_goldens():
    n = 0
    for case, c, exp in golden():
        got = run_hip(c)
        for i in range(N_JOBS):
            same(got[i], exp[i], c["idc"], (case[0], i))
            assert int(got[i][0]["pred_cnt"][0]) >= 1
            n += 1
    assert n == len(CASES) * N_JOBS


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hip_intra_matches_oracle_on_fresh_cases(seed):
    r = np.random.default_rng(seed)
    for k in range(6):
        lw = int(r.integers(2, 7))
        st = int(r.choice([0, 1, 2]))
        case = (5000 + 10 * seed + k, 128, 128 if lw == 6 else 96, int(r.choice([8, 10])), int(r.choice([1, 1, 0, 3])), st, lw, int(r.integers(0, 2)), int(r.integers(0, 2)))
        c = make_case(*case)
        got = run_hip(c)
        for i in range(N_JOBS):
            exp = run_oracle(c, i)
            same(got[i], exp, c["idc"], (case, i))
            assert int(got[i][0]["pred_cnt"][0]) == int(exp[0]["pred_cnt"][0])
