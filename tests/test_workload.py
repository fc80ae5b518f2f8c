"""GPU suite: the bench workload (xeve_amd/workload.py) on a small picture, end results checked against the oracle --
so the thing bench.py times is proven to compute the reference's arithmetic, not just to run."""
import numpy as np
import pytest

from _libs import oracle, ptr

pytestmark = pytest.mark.gpu


def test_hot_path_pass_small_picture_vs_oracle():
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd.workload import N_LIST, PAD_C, PAD_L, HotPathPass

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wl = HotPathPass(256, 128, dev, seed=11)
    wl.run(time_sad=True)
    torch.cuda.synchronize()
    assert set(wl.sad_time_ms()) == set(wl.sizes)
    import ctypes as C

    from _libs import RdoqEst, oracle_rdoq
    from xeve_amd import workload as W

    O, OR = oracle(), oracle_rdoq()
    est = RdoqEst.from_buffer_copy(bytes(wl.rdoq_est))
    bd, qp, s_l, s_c = wl.bd, wl.qp, wl.s_l, wl.s_c
    qs, dqs = D.QUANT_SCALE[0][qp % 6], D.DQ_SCALE[qp % 6] << (qp // 6)
    org = [p.cpu().numpy() for p in wl.org]
    ref = [[p.cpu().numpy() for p in l] for l in wl.ref]
    r = np.random.default_rng(0)
    rate = {S: b.cpu().numpy() for S, b in wl.rate().items()}
    torch.cuda.synchronize()
    from _libs import CU_BITS_JOB_DTYPE, SBAC_DTYPE, CuBitsParams, oracle_sbac

    OS = oracle_sbac()
    for S in wl.sizes:
        lv = wl.lv[S]
        Sc = S // 2
        # F: CABAC bit counts of the quantised CUs (the eight jobs per CU of pinter_residue_rdo's rate term)
        rt = lv["rate"]
        rjobs = rt["jobs"].cpu().numpy().view(CU_BITS_JOB_DTYPE).reshape(lv["n"], wl.RATE_JOBS)
        rcoef, rstate = lv["coef_flat"].cpu().numpy(), rt["state"].cpu().numpy().view(SBAC_DTYPE)
        rp = CuBitsParams.from_buffer_copy(bytes(rt["params"]))
        nnz_dev = np.stack([t.cpu().numpy() for t in lv["nnz"]], axis=1)
        for j in r.choice(lv["n"], size=min(6, lv["n"]), replace=False):
            assert np.array_equal(rjobs["nnz"][j, 1], nnz_dev[j]) and not rjobs["nnz"][j, 0].any()
            for v in range(wl.RATE_JOBS):
                assert rate[S][j, v] == OS.xo_cu_bits(ptr(rstate), None, rp, ptr(rjobs[j, v:v + 1].copy()), ptr(rcoef)), (S, j, v)
        # A: the last integer-search round left its SADs in sad_out
        jobs = lv["me_jobs"][-1].cpu().numpy()
        li = (len(lv["me_jobs"]) - 1) % N_LIST
        got = lv["sad_out"].cpu().numpy()
        cand = wl.cand_l.cpu().numpy()
        for j in r.choice(lv["n"], size=min(6, lv["n"]), replace=False):
            for c in range(len(cand)):
                assert got[j, c] == O.xo_sad(S, S, ptr(org[0], jobs[j, 0]), ptr(ref[li][0], jobs[j, 1] + int(cand[c])), s_l, s_l, bd)
        # D: reconstruction planes = clip(IDCT(dequant(quant(DCT(org - bipred)))) + bipred), for Y, U, V
        rec = [p.cpu().numpy() for p in lv["rec"]]
        fj = [[t.cpu().numpy() for t in pair] for pair in lv["final_jobs"]]
        for j in r.choice(lv["n"], size=min(6, lv["n"]), replace=False):
            for c in range(3):
                w, lg, st = (S, S.bit_length() - 1, s_l) if c == 0 else (Sc, S.bit_length() - 2, s_c)
                pr = []
                for l in range(N_LIST):
                    gx, gy, _, frac = (int(v) for v in fj[l][0 if c == 0 else 1][j])
                    p = np.zeros((w, w), np.int16)
                    (O.xo_mc_l if c == 0 else O.xo_mc_c)(frac & 1, frac >> 1, ptr(ref[l][c]), gx, gy, st, w, ptr(p), w, w, bd,
                                                          O.mc_l_coeff if c == 0 else O.mc_c_coeff)
                    pr.append(p)
                pred = np.zeros((w, w), np.int16)
                O.xo_avg(ptr(pr[0]), ptr(pr[1]), ptr(pred), w, w, w, w, w)
                off = int((lv["off_l"] if c == 0 else lv["off_c"])[j])
                coef = np.zeros(w * w, np.int16)
                O.xo_diff(w, w, ptr(org[c], off), ptr(pred), st, w, w, ptr(coef))
                O.xo_trans(ptr(coef), lg, lg, bd)
                if O.xo_rdoq_zero_test(ptr(coef), lg, lg, qp, qs, 0, bd):
                    if W.USE_RDOQ:
                        OR.xo_rdoq(ptr(coef), lg, lg, qp, wl.lam, int(c == 0), bd, 0, C.byref(est))
                    else:
                        O.xo_quant(ptr(coef), lg, lg, qp, qs, 0, bd)
                else:
                    coef[:] = 0
                O.xo_dquant(ptr(coef), lg, lg, dqs, bd)
                O.xo_itrans(ptr(coef), lg, lg, bd)
                e = np.zeros((w, w), np.int16)
                O.xo_recon(ptr(coef), ptr(pred), 1, w, w, w, ptr(e), bd)
                y0, x0 = off // st, off % st
                assert np.array_equal(rec[c][y0:y0 + w, x0:x0 + w], e), (S, j, c)


def test_structured_content_rdo_phase_vs_oracle():
    """phase G (xeve_hip_residue_rdo_jobs on the structured picture, true-motion bi-prediction) against the oracle's pinter_residue_rdo"""
    import torch

    import xeve_amd
    from _libs import RDO_JOB_DTYPE, RDO_RESULT_DTYPE, SBAC_DTYPE, RdoParams, oracle_rdo
    from _libs import REFPIC_DTYPE
    from xeve_amd.workload import PAD_C, PAD_L, HotPathPass

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wl = HotPathPass(256, 128, dev, seed=3, content="structured")
    out = wl.rdo()
    torch.cuda.synchronize()
    O = oracle_rdo()
    org = [p.cpu().numpy() for p in wl.org]
    ref = [[p.cpu().numpy() for p in l] for l in wl.ref]
    ol, oc = PAD_L * wl.s_l + PAD_L, PAD_C * wl.s_c + PAD_C
    tab = np.zeros(2, REFPIC_DTYPE)
    for l in range(2):
        tab["y"][l], tab["u"][l], tab["v"][l] = ref[l][0].ctypes.data + 2 * ol, ref[l][1].ctypes.data + 2 * oc, ref[l][2].ctypes.data + 2 * oc
        tab["poc"][l] = 2 * l
    org_ptrs = np.array([org[0].ctypes.data + 2 * ol, org[1].ctypes.data + 2 * oc, org[2].ctypes.data + 2 * oc], np.uint64)
    r = np.random.default_rng(1)
    coded = 0
    for S in wl.sizes:
        rd = wl.lv[S]["rdo"]
        res = out[S][0].cpu().numpy().reshape(-1).view(RDO_RESULT_DTYPE)
        jobs = rd["jobs"].cpu().numpy().view(RDO_JOB_DTYPE)
        st = rd["state"].cpu().numpy().view(SBAC_DTYPE)
        p = RdoParams.from_buffer_copy(bytes(rd["params"]))
        for j in r.choice(len(jobs), size=min(8, len(jobs)), replace=False):
            er, eb = np.zeros(1, RDO_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ec = [np.zeros(S * S, np.int16), np.zeros(S * S // 4, np.int16), np.zeros(S * S // 4, np.int16)]
            O.xo_residue_rdo(ptr(org_ptrs), wl.s_l, wl.s_c, ptr(tab), wl.s_l, wl.s_c, ptr(st), p, ptr(jobs[j:j + 1].copy()), ptr(er), ptr(ec[0]), ptr(ec[1]),
                             ptr(ec[2]), ptr(eb))
            assert res["cost"][j].tobytes() == er["cost"][0].tobytes() and np.array_equal(res["nnz"][j], er["nnz"][0]), (S, j)
            coded += int(er["nnz"][0].any())
    assert coded > 0


@pytest.mark.gpu
@pytest.mark.parametrize("content", ["structured", "iid"])
def test_inter_phase_vs_oracle(content):
    """phase H = the step bench.py times (xeve_hip_pinter_analyze_cu_jobs: the whole inter analysis of every CU of every level), on the structured and on the
    i.i.d. picture, against the oracle"""
    import torch

    import xeve_amd
    from _inter_cases import oracle_params_from_hip
    from _libs import INTER_JOB_DTYPE, INTER_RESULT_DTYPE, REFPIC_DTYPE, SBAC_DTYPE, oracle_inter
    from xeve_amd.workload import PAD_C, PAD_L, HotPathPass

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wl = HotPathPass(256, 128, dev, seed=3, content=content)
    out = wl.inter()
    torch.cuda.synchronize()
    O = oracle_inter()
    org = [p.cpu().numpy() for p in wl.org]
    ref = [[p.cpu().numpy() for p in l] for l in wl.ref]
    ol, oc = PAD_L * wl.s_l + PAD_L, PAD_C * wl.s_c + PAD_C
    org_ptrs = np.array([org[0].ctypes.data + 2 * ol, org[1].ctypes.data + 2 * oc, org[2].ctypes.data + 2 * oc], np.uint64)
    r = np.random.default_rng(2)
    modes = set()
    for S in wl.sizes:
        h = wl.lv[S]["inter"]
        tab = np.zeros(2, REFPIC_DTYPE)
        for l in range(2):
            tab["y"][l], tab["u"][l], tab["v"][l] = ref[l][0].ctypes.data + 2 * ol, ref[l][1].ctypes.data + 2 * oc, ref[l][2].ctypes.data + 2 * oc
            tab["poc"][l] = int(h["refp"]["poc"][l])
        res = out[S].cpu().numpy().reshape(-1).view(INTER_RESULT_DTYPE)
        jobs = h["jobs"].cpu().numpy().view(INTER_JOB_DTYPE)
        st = wl.lv[S]["rdo"]["state"].cpu().numpy().view(SBAC_DTYPE)
        P = oracle_params_from_hip(h["params"])
        for j in r.choice(len(jobs), size=min(6, len(jobs)), replace=False):
            er, eb = np.zeros(1, INTER_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ec = [np.zeros(S * S, np.int16), np.zeros(S * S // 4, np.int16), np.zeros(S * S // 4, np.int16)]
            ep = [x.copy() for x in ec]
            O.xo_pinter_analyze_cu(ptr(org_ptrs), wl.s_l, wl.s_c, ptr(tab), wl.s_l, wl.s_c, ptr(st), P, ptr(jobs[j:j + 1].copy()), ptr(er), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]),
                                   ptr(ep[0]), ptr(ep[1]), ptr(ep[2]), ptr(eb))
            assert res[j:j + 1].tobytes() == er.tobytes(), (S, j, res[j], er[0])
            modes.add(int(er["best_idx"][0]))
    assert len(modes) >= (2 if content == "structured" else 1), modes


@pytest.mark.gpu
@pytest.mark.parametrize("content", ["structured", "iid"])
def test_intra_phase_vs_oracle(content):
    """phase I (xeve_hip_pintra_analyze_cu_jobs on every CU of every level 64 .. 4, neighbours from reference picture 0) against the oracle on sampled CUs"""
    import ctypes as C

    import torch

    import xeve_amd
    from _intra_cases import INTRA_JOB_DTYPE, INTRA_RESULT_DTYPE, IntraParams, oracle_intra
    from _libs import SBAC_DTYPE
    from xeve_amd.workload import PAD_C, PAD_L, HotPathPass

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wl = HotPathPass(256, 128, dev, seed=3, content=content)
    out = wl.intra()
    torch.cuda.synchronize()
    O = oracle_intra()
    org = [p.cpu().numpy() for p in wl.org]
    mod = [p.cpu().numpy() for p in wl.ref[0]]
    ol, oc = PAD_L * wl.s_l + PAD_L, PAD_C * wl.s_c + PAD_C
    optr = (C.c_void_p * 3)(org[0].ctypes.data + 2 * ol, org[1].ctypes.data + 2 * oc, org[2].ctypes.data + 2 * oc)
    mptr = (C.c_void_p * 3)(mod[0].ctypes.data + 2 * ol, mod[1].ctypes.data + 2 * oc, mod[2].ctypes.data + 2 * oc)
    m = wl._intra_maps
    scu, ipm, tidx = m["scu"].cpu().numpy().view(np.uint32), m["ipm"].cpu().numpy(), m["tidx"].cpu().numpy()
    st = m["state"].cpu().numpy().view(SBAC_DTYPE)
    r = np.random.default_rng(4)
    modes = set()
    for S in wl.INTRA_SIZES:
        h = wl._ilv[S]
        res, coef, rec, best = (t.cpu().numpy() for t in out[S])
        res, best = res.reshape(-1).view(INTRA_RESULT_DTYPE), best.reshape(-1).view(SBAC_DTYPE)
        jobs = h["jobs"].cpu().numpy().view(INTRA_JOB_DTYPE)
        P = IntraParams.from_buffer_copy(bytes(h["params"]))
        n, n0, n1 = len(jobs), S * S, S * S // 4
        for i in [0, n - 1] + [int(v) for v in r.integers(0, n, size=6)]:
            er, eb = np.zeros(1, INTRA_RESULT_DTYPE), np.zeros(1, SBAC_DTYPE)
            ec, ek = [np.zeros(n0, np.int16), np.zeros(n1, np.int16), np.zeros(n1, np.int16)], [np.zeros(n0, np.int16), np.zeros(n1, np.int16), np.zeros(n1, np.int16)]
            O.xo_pintra_analyze_cu(optr, wl.s_l, wl.s_c, mptr, wl.s_l, wl.s_c, ptr(scu), ptr(ipm), ptr(tidx), ptr(st), C.byref(P), ptr(jobs[i:i + 1]), ptr(er), ptr(ec[0]),
                                   ptr(ec[1]), ptr(ec[2]), ptr(ek[0]), ptr(ek[1]), ptr(ek[2]), ptr(eb))
            assert res[i:i + 1].tobytes() == er.tobytes(), (S, i, res[i], er[0])
            assert best[i:i + 1].tobytes() == eb.tobytes(), (S, i)
            for c, (base, nn) in enumerate(((i * n0, n0), (n * n0 + i * n1, n1), (n * (n0 + n1) + i * n1, n1))):
                assert np.array_equal(coef[base:base + nn], ec[c]) and np.array_equal(rec[base:base + nn], ek[c]), (S, i, c)
            modes.add(int(er["ipm"][0, 0]))
    assert len(modes) >= 3
