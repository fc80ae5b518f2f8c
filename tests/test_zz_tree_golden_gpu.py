"""GPU: the device-side CTU mode decision against the golden fixture recorded from the reference encoder (tests/golden/tree_v1.npz).  (The file sorts last on purpose:
this test was written after the round's GPU budget was spent -- its plumbing is checked on the CPU with the oracle as the engine, tests/test_tree_golden.py -- and a
`pytest -x` run should reach it after everything else.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_hip_ctu_mode_decision_matches_the_reference_goldens(each_walk):
    """the device walk against tests/golden/tree_v1.npz: CTUs of real encodes (I, P and B pictures, intra / inter / skip / direct CUs) with what the REFERENCE made of them,
    recorded inside the unmodified encoder (tests/golden/make_tree_golden.py) -- no oracle in between"""
    import torch
    import xeve_amd
    from xeve_amd import device as D
    from _tree_golden import load, run_walk, same_as_reference

    xeve_amd.init(0)
    n = 0
    for r in load():
        same_as_reference(r, *run_walk(r, torch.device("cuda:0"), D.mode_analyze_ctu_jobs))
        n += 1
    assert n >= 10


def test_hip_writer_matches_the_reference_goldens():
    """xeve_hip_eco_ctu_jobs against the reference WRITER's side of the recorded CTUs: the coder state xeve_eco_tree left, the bytes it put into the bitstream, the unit flags"""
    import numpy as np
    import torch
    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib
    from _libs import SBAC_DTYPE
    from _tree_golden import load, writer_inputs, writer_same_as_reference

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    n = 0
    for r in load():
        if "wr" not in r:
            continue
        d, st, m = writer_inputs(r)
        EP = lib.EcoParams()
        EP.chroma_format_idc, EP.slice_type, EP.log2_ctu, EP.pic_w, EP.pic_h, EP.w_scu, EP.h_scu = r["idc"], r["slice_type"], r["P"].log2_ctu, r["w"], r["h"], r["w"] // 4, r["h"] // 4
        EP.num_refp[0], EP.num_refp[1] = r["wr"]["num_refp"]
        jobs = np.zeros(1, np.dtype(lib.CTU_JOB_DTYPE))
        jobs["x"], jobs["y"] = r["x0"], r["y0"]
        ctus = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
        states = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
        ms, mc = (torch.from_numpy(m[k].view(np.int32).copy()).to(dev) for k in ("scu", "cu_mode"))
        mi, mt = (torch.from_numpy(m[k].copy()).to(dev) for k in ("ipm", "tidx"))
        by, nb = D.eco_ctu_jobs(ctus, states, EP, ms, mi, mt, mc, torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), bytes_cap=1 << 16)
        torch.cuda.synchronize()
        k = int(nb.cpu().numpy()[0])
        writer_same_as_reference(r, states.cpu().numpy().reshape(-1).view(SBAC_DTYPE), by.cpu().numpy()[0][:k],
                                 dict(scu=ms.cpu().numpy().view(np.uint32), cu_mode=mc.cpu().numpy().view(np.uint32)))
        n += 1
    assert n >= 8
