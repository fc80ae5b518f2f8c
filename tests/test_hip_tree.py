"""The mode decision of I-picture CTUs on the GPU (xeve_hip_mode_analyze_ctu_intra_jobs) against the pinned oracle: whole small pictures coded CTU by CTU, the
pictures of a case as the chains of one call.  After every CTU: the CTU's data (split modes, prediction modes, depths, nnz, map fields, levels, reconstruction)
byte for byte, the coder state handed to the next CTU, the cost as the bit pattern of the double; at the end the reconstructed pictures and the 4x4-unit maps."""
import numpy as np
import pytest

from _libs import SBAC_DTYPE
from _tree_cases import CASES, CTU_DATA_DTYPE, CTU_JOB_DTYPE, make_case, run_oracle_picture

pytestmark = pytest.mark.gpu


def run_hip_case(c):
    import torch
    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    n = c["npic"]
    org = [torch.from_numpy(a.copy()).to(dev) for a in c["org"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in c["mod"]]
    m = c["maps"]
    ms, mi, mt, mc = (torch.from_numpy(m[k].view(np.int32 if m[k].dtype == np.uint32 else m[k].dtype).copy()).to(dev) for k in ("scu", "ipm", "tidx", "cu_mode"))
    P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
    states = torch.from_numpy(c["entry"].view(np.uint8).copy()).to(dev)
    pe = (org[0][0].numel(), org[1][0].numel(), mod[0][0].numel(), mod[1][0].numel(), m["scu"].shape[1])
    per_ctu = []
    for (x, y) in c["order"]:
        jobs = np.zeros(n, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"], jobs["sbac"], jobs["pic"] = x, y, np.arange(n), np.arange(n)
        out, nxt, cost = D.mode_analyze_ctu_intra_jobs([t.data_ptr() for t in org], org[0].shape[2], org[1].shape[2], [t.data_ptr() for t in mod], mod[0].shape[2],
                                                       mod[1].shape[2], ms, mi, mt, mc, states, P, torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), pic_elems=pe)
        torch.cuda.synchronize()
        per_ctu.append((out.cpu().numpy().reshape(-1).view(CTU_DATA_DTYPE), nxt.cpu().numpy().reshape(-1).view(SBAC_DTYPE), cost.cpu().numpy()))
        states = nxt.clone()
    final = dict(mod=[t.cpu().numpy() for t in mod], scu=ms.cpu().numpy().view(np.uint32), ipm=mi.cpu().numpy(), cu_mode=mc.cpu().numpy().view(np.uint32))
    return per_ctu, final


def compare(case, c, got, final):
    exp = [run_oracle_picture(c, p) for p in range(c["npic"])]  # updates c["mod"], c["maps"] in place
    for k in range(len(c["order"])):
        d, nb, cost = got[k]
        for p in range(c["npic"]):
            ed, enb, ecost = exp[p][k]
            for f in CTU_DATA_DTYPE.names:
                assert np.array_equal(d[f][p], ed[f][0]), (case, "ctu", k, "picture", p, f)
            assert nb[p:p + 1].tobytes() == enb.tobytes(), (case, k, p, "coder state")
            assert np.float64(cost[p]).tobytes() == np.float64(ecost).tobytes(), (case, k, p, cost[p], ecost)
    for j in range(3 if c["idc"] else 1):
        assert np.array_equal(final["mod"][j], c["mod"][j]), (case, "picture", j)
    for f in ("scu", "ipm", "cu_mode"):
        assert np.array_equal(final[f].reshape(c["maps"][f].shape), c["maps"][f]), (case, "map", f)


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_hip_ctu_mode_decision_matches_oracle(case):
    c = make_case(*case)
    got, final = run_hip_case(c)
    compare(case, c, got, final)
    # the walk did decide something: some CTU is split and some CU is kept whole
    depths = np.concatenate([g[0]["depth"].reshape(-1) for g in got])
    assert len(np.unique(depths)) >= 2
