"""GPU suite: xeve_hip_me_ipel_diamond_jobs (one complete me_ipel_diamond per wave) against the reference goldens and
against the oracle on batches of seeded jobs."""
import ctypes as C

import numpy as np
import pytest

from _me_cases import PAD, make_job, make_planes, run_oracle
from _me_golden import golden_cases

pytestmark = pytest.mark.gpu


def gpu_run(cases):
    """all cases share planes, block size and every launch-level parameter"""
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    c0 = cases[0]
    S = c0["S"]
    jobs = np.zeros(len(cases), dtype=lib.ME_JOB_DTYPE)
    bi_buf = np.zeros((len(cases), S * S), np.int16)
    for i, c in enumerate(cases):
        jobs[i] = (c["x"], c["y"], i * S * S, c["range"], c["gmvp"], c["mvi"], c["beststep_in"])
        bi_buf[i] = c["org_bi"]
    P = lib.MeParams(c0["lambda_mv"], 1, c0["mot_other"], c0["bi"], c0["faststep"], c0["msr"], c0["sr"], (C.c_int32 * 2)(*c0["min_clip"]),
                     (C.c_int32 * 2)(*c0["max_clip"]), 0)
    org, ref = torch.from_numpy(c0["org"]).to(dev), torch.from_numpy(c0["ref"]).to(dev)
    o0 = PAD * c0["s"] + PAD
    return D.me_ipel_diamond_jobs(org, o0, c0["s"], torch.from_numpy(bi_buf).to(dev), ref, o0, c0["s"], jobs, S.bit_length() - 1, 10, P)


def test_me_matches_reference_goldens():
    n = 0
    for c, (cost, mvx, mvy, beststep) in golden_cases():
        res = gpu_run([c])[0]
        assert (int(res["cost"]), int(res["mv"][0]), int(res["mv"][1]), int(res["beststep"])) == (cost, mvx, mvy, beststep), (n, c["S"], c["bi"])
        n += 1
    assert n == 96


@pytest.mark.parametrize("S", [8, 16, 32, 64])
@pytest.mark.parametrize("bi", [0, 1, 2])
@pytest.mark.parametrize("textured", [False, True])
def test_me_batches_vs_oracle(S, bi, textured):
    r = np.random.default_rng(1000 + S + bi * 7 + textured)
    pl = make_planes(r, textured)
    base = make_job(r, pl, S, bi)
    cases = []
    for _ in range(150):
        c = make_job(r, pl, S, bi)
        for k in ("lambda_mv", "mot_other", "faststep", "msr", "sr"):  # launch-level parameters are shared
            c[k] = base[k]
        # the job's own range must be derived with the shared search range
        sr = base["sr"]
        cx = min(max(c["x"] + ((c["gmvp"][0] - (c["x"] << 2)) >> 2), c["min_clip"][0]), c["max_clip"][0])
        cy = min(max(c["y"] + ((c["gmvp"][1] - (c["y"] << 2)) >> 2), c["min_clip"][1]), c["max_clip"][1])
        c["range"] = [min(max(cx - sr, c["min_clip"][0]), c["max_clip"][0]), min(max(cy - sr, c["min_clip"][1]), c["max_clip"][1]),
                      min(max(cx + sr, c["min_clip"][0]), c["max_clip"][0]), min(max(cy + sr, c["min_clip"][1]), c["max_clip"][1])]
        cases.append(c)
    got = gpu_run(cases)
    steps = set()
    for i, c in enumerate(cases):
        e = run_oracle(c)
        g = got[i]
        assert (int(g["cost"]), int(g["mv"][0]), int(g["mv"][1]), int(g["beststep"]), int(g["best_mv_bits"])) == \
               (e.cost, e.mv[0], e.mv[1], e.beststep, e.best_mv_bits), (S, bi, textured, i)
        steps.add(e.beststep)
    assert len(steps) >= 1


def gpu_run_spel(cases):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    c0 = cases[0]
    S = c0["S"]
    jobs = np.zeros(len(cases), dtype=lib.SPEL_JOB_DTYPE)
    bi_buf = np.zeros((len(cases), S * S), np.int16)
    for i, c in enumerate(cases):
        jobs[i] = (c["x"], c["y"], i * S * S, c["gmvp"], c["mvi"])
        bi_buf[i] = c["org_bi"]
    P = lib.SpelParams(c0["lambda_mv"], 1, c0["mot_other"], c0["bi"], c0["hpel_cnt"], c0["qpel_cnt"])
    org, ref = torch.from_numpy(c0["org"]).to(dev), torch.from_numpy(c0["ref"]).to(dev)
    o0 = PAD * c0["s"] + PAD
    return D.me_spel_pattern_jobs(org, o0, c0["s"], torch.from_numpy(bi_buf).to(dev), ref, o0, c0["s"], jobs, S.bit_length() - 1, 10, P)


def test_spel_matches_reference_goldens():
    from _me_golden import golden_spel_cases

    n = 0
    for c, (cost, mvx, mvy) in golden_spel_cases():
        res = gpu_run_spel([c])[0]
        assert (int(res["cost"]), int(res["mv"][0]), int(res["mv"][1])) == (cost, mvx, mvy), (n, c["S"], c["bi"])
        n += 1
    assert n == 64


@pytest.mark.parametrize("S", [8, 16, 32, 64])
@pytest.mark.parametrize("bi", [0, 1])
def test_spel_batches_vs_oracle(S, bi):
    from _me_cases import make_spel_job, run_oracle_spel

    r = np.random.default_rng(1100 + S + bi)
    pl = make_planes(r, True)
    base = make_spel_job(r, pl, S, bi)
    cases = []
    for _ in range(120):
        c = make_spel_job(r, pl, S, bi)
        for k in ("lambda_mv", "mot_other", "hpel_cnt", "qpel_cnt"):
            c[k] = base[k]
        cases.append(c)
    got = gpu_run_spel(cases)
    for i, c in enumerate(cases):
        e = run_oracle_spel(c)
        g = got[i]
        assert (int(g["cost"]), int(g["mv"][0]), int(g["mv"][1]), int(g["best_mv_bits"])) == (e.cost, e.mv[0], e.mv[1], e.best_mv_bits), (S, bi, i)


@pytest.mark.parametrize("S", [8, 16, 32, 64])
@pytest.mark.parametrize("bi", [0, 1])
def test_epzs_search_vs_oracle(S, bi):
    """xeve_amd.me.epzs_search_device = xeve_hip_me_epzs_jobs: pinter_me_epzs per block (diamond, refinement loop, sub-pel) on the GPU, vs the oracle"""
    import torch

    import xeve_amd
    from _me_cases import make_epzs_job, run_oracle_epzs
    from xeve_amd import me

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    r = np.random.default_rng(1200 + S + bi)
    pl = make_planes(r, True)
    base = make_epzs_job(r, pl, S, bi)
    cases = []
    for _ in range(100):
        c = make_epzs_job(r, pl, S, bi)
        for k in ("lambda_mv", "mot_other", "msr", "sr", "hpel_cnt", "qpel_cnt"):
            c[k] = base[k]
        cases.append(c)
    org, ref = torch.from_numpy(pl["org"]).to(dev), torch.from_numpy(pl["ref"]).to(dev)
    o0 = PAD * pl["s"] + PAD
    org_bi = torch.from_numpy(np.stack([c["org_bi"] for c in cases])).to(dev)
    args = (org, o0, pl["s"], ref, o0, pl["s"], [c["x"] for c in cases], [c["y"] for c in cases], [c["mvp"] for c in cases],
            S.bit_length() - 1, 10, base["lambda_mv"], 1, base["msr"], base["sr"], base["min_clip"], base["max_clip"], base["hpel_cnt"], base["qpel_cnt"])
    kw = dict(bi=bi, org_bi=org_bi, mv_start=[c["mv0"] for c in cases], extra_bits=base["mot_other"])
    exp = [run_oracle_epzs(c) for c in cases]
    cost, mv = me.epzs_search_device(*args, **kw)
    for i in range(len(cases)):
        assert (int(cost[i]), int(mv[i, 0]), int(mv[i, 1])) == exp[i], (S, bi, i)
    # the side effect on pi->mot_bits[lidx] (the next bi-directional search reads it): oracle -1 = untouched = 0 here
    cost, mv, mot = me.epzs_search_device(*args, with_mot_bits=True, **kw)
    for i, c in enumerate(cases):
        e = run_oracle_epzs(c, with_mot=True)
        assert int(mot[i]) == (e[3] if e[3] > 0 else 0), (S, bi, i)


@pytest.mark.parametrize("S", [8, 16, 32, 64])
def test_hip_epzs_raster_search_and_integer_refinement_vs_oracle(S):
    """the branches of pinter_me_epzs outside presets fast / medium: me_raster (me_complexity > 1) after a first search that ended far from its start,
    with refi 0 / 1, and me_ipel_refinement instead of the sub-pel pattern (me_level = ME_LEV_IPEL), uni- and bi-directional"""
    import torch

    import xeve_amd
    from _me_cases import make_epzs_job, run_oracle_epzs
    from xeve_amd import me

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    r = np.random.default_rng(3100 + S)
    pl = make_planes(r, True)
    org, ref = torch.from_numpy(pl["org"]).to(dev), torch.from_numpy(pl["ref"]).to(dev)
    o0 = PAD * pl["s"] + PAD
    rastered = 0
    for (bi, raster, refi, ipel) in [(0, 1, 0, 0), (0, 1, 1, 0), (0, 1, 1, 1), (0, 0, 0, 1), (1, 1, 0, 1), (1, 0, 0, 0)]:
        base = make_epzs_job(r, pl, S, bi)
        if ipel:
            base["hpel_cnt"], base["qpel_cnt"] = 0, 0
        cases = []
        for _ in range(60):
            c = make_epzs_job(r, pl, S, bi)
            for k in ("lambda_mv", "mot_other", "msr", "sr", "hpel_cnt", "qpel_cnt"):
                c[k] = base[k]
            c["mvp"] = (int(r.integers(-160, 161)), int(r.integers(-160, 161)))
            c["raster"], c["refi"] = raster, refi
            cases.append(c)
        org_bi = torch.from_numpy(np.stack([c["org_bi"] for c in cases])).to(dev)
        cost, mv, mot = me.epzs_search_device(org, o0, pl["s"], ref, o0, pl["s"], [c["x"] for c in cases], [c["y"] for c in cases], [c["mvp"] for c in cases],
                                              S.bit_length() - 1, 10, base["lambda_mv"], 1, base["msr"], base["sr"], base["min_clip"], base["max_clip"], base["hpel_cnt"],
                                              base["qpel_cnt"], bi=bi, org_bi=org_bi, mv_start=[c["mv0"] for c in cases], extra_bits=base["mot_other"], with_mot_bits=True,
                                              raster=bool(raster), refi=refi)
        for i, c in enumerate(cases):
            e = run_oracle_epzs(c, with_mot=True)
            assert (int(cost[i]), int(mv[i, 0]), int(mv[i, 1]), int(mot[i])) == (e[0], e[1], e[2], e[3] if e[3] > 0 else 0), (S, bi, raster, refi, ipel, i, c["mvp"])
            if raster and bi == 0:
                rastered += run_oracle_epzs(dict(c, raster=0), with_mot=True) != e
    assert rastered > 0
