"""The fused CTU walk (xeve_amd/csrc/walk.h: what libxeve_hip.so runs as one kernel per CTU step) against the pinned oracle, on the CPU: the header's functions are
__host__ __device__ and tests/native builds their host side as a team of one thread.  Whole small pictures coded CTU by CTU, the pictures of a case as the chains of one
call (several chains per team: the team-local lockstep is exercised too).  After every CTU: the CTU's data byte for byte, the coder state handed on field for field,
the cost as the bit pattern of the double; at the end the reconstructed pictures and the 4x4-unit maps."""
import os

import numpy as np
import pytest

import _walk
from _libs import SBAC_DTYPE
from _tree_cases import CASES, CTU_DATA_DTYPE, CTU_JOB_DTYPE, make_case, run_oracle_picture

pytestmark = pytest.mark.skipif(not _walk.available(), reason="hipcc not found")


def run_walk_case(c, chains_per_team, full=1, threads=1):
    n = c["npic"]
    org = [a.copy() for a in c["org"]]
    mod = [a.copy() for a in c["mod"]]
    m = {k: v.copy() for k, v in c["maps"].items()}
    states = c["entry"].copy()
    pe = (org[0][0].size, org[1][0].size, mod[0][0].size, mod[1][0].size, m["scu"].shape[1])
    per_ctu = []
    for (x, y) in c["order"]:
        jobs = np.zeros(n, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"], jobs["sbac"], jobs["pic"] = x, y, np.arange(n), np.arange(n)
        out, nxt, cost = _walk.host_walk([a.ctypes.data for a in org], org[0].shape[2], org[1].shape[2], [a.ctypes.data for a in mod], mod[0].shape[2], mod[1].shape[2],
                                         m["scu"], m["ipm"], m["tidx"], m["cu_mode"], pe, states, c["P"], None, jobs, chains_per_team, full, 0, threads)
        per_ctu.append((out, nxt, cost))
        states = nxt.copy()
    return per_ctu, dict(mod=mod, scu=m["scu"], ipm=m["ipm"], cu_mode=m["cu_mode"])


def compare(case, c, got, final, full=1):
    exp = [run_oracle_picture(c, p) for p in range(c["npic"])]  # updates c["mod"], c["maps"] in place
    for k in range(len(c["order"])):
        d, nb, cost = got[k]
        for p in range(c["npic"]):
            ed, enb, ecost = exp[p][k]
            for f in CTU_DATA_DTYPE.names:
                assert np.array_equal(d[f][p], ed[f][0]), (case, "ctu", k, "picture", p, f)
            if full:
                assert nb[p:p + 1].tobytes() == enb.tobytes(), (case, k, p, "coder state")
            else:  # count-only states: what a later count depends on -- the range and the models
                assert int(nb["range"][p]) == int(enb["range"][0]) and np.array_equal(nb["ctx"][p], enb["ctx"][0]), (case, k, p, "count-only coder state")
            assert np.float64(cost[p]).tobytes() == np.float64(ecost).tobytes(), (case, k, p, cost[p], ecost)
    for j in range(3 if c["idc"] else 1):
        assert np.array_equal(final["mod"][j], c["mod"][j]), (case, "picture", j)
    for f in ("scu", "ipm", "cu_mode"):
        assert np.array_equal(final[f].reshape(c["maps"][f].shape), c["maps"][f]), (case, "map", f)


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_walk_i_pictures_match_oracle(case):
    c = make_case(*case)
    got, final = run_walk_case(c, chains_per_team=2)
    compare(case, c, got, final)


@pytest.mark.parametrize("case", CASES[:3], ids=[str(c[0]) for c in CASES[:3]])
def test_walk_i_pictures_with_count_only_states(case):
    """the encoder's form (the walk's exit states are only ever loaded into further counts): the same decisions and costs, the states carry range + models"""
    c = make_case(*case)
    got, final = run_walk_case(c, chains_per_team=3, full=0)
    compare(case, c, got, final, full=0)


# ---- P / B slices ------------------------------------------------------------------------------------------------------------------------------------------------
from _tree_cases import INTER_CASES, make_inter_case, run_oracle_inter_picture  # noqa: E402


def run_walk_inter_case(c, full=1, threads=1):
    import ctypes as C

    from _mc_cases import refpic_table
    from test_hip_inter import hip_params
    from xeve_amd import lib
    from xeve_amd.device import baseline_coef_c, baseline_coef_l

    refs, org = c["refs"], c["org"]
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    org_ptrs = [int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"]]
    mod = [a.copy() for a in c["mod"]]
    m = {k: v.copy() for k, v in c["maps"].items()}
    col = [a.copy() for a in c["col"]]
    I = lib.TreeInter()
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, refs["s_l"], refs["s_c"], hip_params(c["ipar"])
    I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, col[0].ctypes.data, col[1].ctypes.data, c["ecu_depth"]
    cl, cc = baseline_coef_l(), baseline_coef_c()
    I.coef_l, I.coef_c = cl.ctypes.data, cc.ctypes.data
    states = c["entry"][0:1].copy()
    per_ctu = []
    for (x, y) in c["order"]:
        jobs = np.zeros(1, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"] = x, y
        out, nxt, cost = _walk.host_walk(org_ptrs, refs["s_l"], refs["s_c"], [a.ctypes.data for a in mod], mod[0].shape[1], mod[1].shape[1], m["scu"], m["ipm"], m["tidx"],
                                         m["cu_mode"], None, states, c["P"], I, jobs, 1, full, 0, threads)
        per_ctu.append((out, nxt, cost))
        states = nxt.copy()
    return per_ctu, dict(mod=mod, scu=m["scu"], ipm=m["ipm"], cu_mode=m["cu_mode"], mv=m["mv"], refi=m["refi"])


def check_inter(case, threads=1):
    c = make_inter_case(*case)
    got, final = run_walk_inter_case(c, threads=threads)
    exp = run_oracle_inter_picture(c)  # updates c["mod"], c["maps"] in place
    for k in range(len(c["order"])):
        d, nb, cost = got[k]
        ed, enb, ecost = exp[k]
        for f in CTU_DATA_DTYPE.names:
            assert np.array_equal(d[f][0], ed[f][0]), (case, "ctu", k, f, np.argwhere(d[f][0] != ed[f][0])[:4].tolist())
        assert nb[0:1].tobytes() == enb.tobytes(), (case, k, "coder state")
        assert np.float64(cost[0]).tobytes() == np.float64(ecost).tobytes(), (case, k, cost[0], ecost)
    for j in range(3 if c["idc"] else 1):
        assert np.array_equal(final["mod"][j], c["mod"][j]), (case, "picture", j)
    for f in ("scu", "ipm", "cu_mode", "mv", "refi"):
        assert np.array_equal(final[f].reshape(c["maps"][f].shape), c["maps"][f]), (case, "map", f)


@pytest.mark.parametrize("case", INTER_CASES, ids=[str(c[0]) for c in INTER_CASES])
def test_walk_p_and_b_pictures_match_oracle(case):
    check_inter(case)


# ---- the device's team as REAL threads, sync() a barrier: the lane mapping of every stage as the kernel runs it (64 lanes = one wave by default; XEVE_RACE_TESTS=1: the whole
# 256, larger pictures -- minutes on 8 cores, 10 000 barriers per CTU).  The results must not depend on the team's size.  tests/test_walk_race.py runs the same team under
# ThreadSanitizer.
SLOW = bool(os.environ.get("XEVE_RACE_TESTS"))


@pytest.mark.parametrize("threads", [64, 256])
def test_walk_i_picture_with_a_team_of_real_threads(threads):
    for case, C, full in ((CASES[0], 2, 1), (CASES[1], 3, 0)) if SLOW else ((CASES[4], 2, 0),):
        c = make_case(*case)
        got, final = run_walk_case(c, chains_per_team=C, full=full, threads=threads)
        compare(case, c, got, final, full=full)


@pytest.mark.parametrize("case", INTER_CASES[:3] if SLOW else INTER_CASES[3:4], ids=[str(c[0]) for c in (INTER_CASES[:3] if SLOW else INTER_CASES[3:4])])
@pytest.mark.parametrize("threads", [64, 256] if SLOW else [64])
def test_walk_p_and_b_pictures_with_a_team_of_real_threads(case, threads):
    check_inter(case, threads=threads)


@pytest.mark.parametrize("rotate", [1, 3] if SLOW else [3])  # (a rotation by one wave is the same code path with another constant: 65 s more, with XEVE_RACE_TESTS=1)
def test_the_serial_stages_packed_into_a_rotated_wave_change_nothing(rotate, monkeypatch):
    """walk.hip runs a team's serial stages (coder jobs, RDOQ scans) on ONE wave and takes that wave from a different SIMD for each of a CU's four teams: the logical thread
    index is the hardware one with the waves rotated.  The same layouts on the host's team of 256 real threads (P::deal = 1, the thread indices rotated by whole waves) must
    give the oracle's results: an I picture with two chains per team and a B picture"""
    monkeypatch.setenv("XW_HOST_DEAL", "1")
    monkeypatch.setenv("XW_HOST_ROTATE", str(rotate))
    c = make_case(*CASES[4])
    got, final = run_walk_case(c, chains_per_team=2, full=0, threads=256)
    compare(CASES[4], c, got, final, full=0)
    check_inter(INTER_CASES[3], threads=256)


# ---- end to end: the batch encoder's frame loop with EVERY CTU decided by the fused walk's host side, against bitstreams of the unmodified reference application ----------
import json  # noqa: E402

import _e2e  # noqa: E402
import _enc  # noqa: E402


@pytest.fixture(scope="module")
def walk_engine():
    h = _enc.harness()
    was = h.xo_encode_use_walk(1)
    yield h
    h.xo_encode_use_walk(was)


def _clip(tmp, name, w, h, n, seed):
    p = os.path.join(tmp, name + ".yuv")
    if not os.path.exists(p):
        _e2e.make_yuv(p, w, h, n, seed)
    return open(p, "rb").read()


@pytest.mark.parametrize("name", sorted(_e2e.CASES))
def test_walk_decides_every_ctu_of_the_single_runs(name, walk_engine, tmp_path_factory):
    """all-intra, low-delay B, random access, closed GOP, two row chains, partial CTUs: count-only coder states, the chains of a wavefront step as one team"""
    E2E = json.load(open(os.path.join(_enc.ROOT, "tests", "golden", "e2e_v1.json")))
    w, h, n, seed, cli = _e2e.CASES[name]
    out = _enc.encode_cpu(_enc.config(w, h, cli), [_clip(str(tmp_path_factory.getbasetemp()), name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.BATCH_CASES))
def test_walk_decides_every_ctu_of_the_closed_gop_batches(name, walk_engine, tmp_path_factory):
    w, h, gops, frames, seed, cli, threads = _enc.BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _clip(str(tmp_path_factory.getbasetemp()), name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


@pytest.mark.parametrize("name", sorted(_enc.HOST_PINNED_CASES))
def test_walk_decides_every_ctu_with_p_slices_and_chroma_qp_offsets(name, walk_engine, tmp_path_factory):
    """--inter-slice-type 1 and --qp-cb-offset / --qp-cr-offset (goldens from the reference LIBRARY, oracle/ref_param_pin.c): P pictures through the fused walk's
    one-list inter analysis, chroma QPs / lambdas / weights of their own -- what the product accepts since round 5"""
    w, h, gops, frames, seed, cli, threads = _enc.HOST_PINNED_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _clip(str(tmp_path_factory.getbasetemp()), name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


@pytest.mark.parametrize("name", ["slow_moving_ra_b3", "slow_cif_closed_gop", "slow_noise_allintra", "slow_jumpy_ldb"])
def test_walk_decides_every_ctu_at_preset_slow(name, walk_engine, tmp_path_factory):
    """--preset slow through the fused walk's host side: the quarter-pel stage of its search and walk_dbk.h -- the loop filter's share of every candidate's distortion
    (merge pairs, the inter candidates' prediction and reconstruction, the intra candidates' luma and chroma) -- against the reference application's bitstreams"""
    E2E = json.load(open(os.path.join(_enc.ROOT, "tests", "golden", "e2e_v1.json")))
    w, h, n, seed, cli = _e2e.SLOW_CASES[name]
    out = _enc.encode_cpu(_enc.config(w, h, cli), [_clip(str(tmp_path_factory.getbasetemp()), name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_e2e.PLACEBO_CASES))
def test_walk_decides_every_ctu_at_preset_placebo(name, walk_engine, tmp_path_factory):
    """--preset placebo through the fused walk's host side: inter CUs of 4x4 (the search's 4-sample rows, 2x2 chroma blocks, the 4x4 Hadamard of the winner), two reference
    pictures per list, the raster search (CT_GRID rounds), the diamond's rings up to 256 of a range of 384, 64x64 intra CUs in I slices, four merge candidates = 16 pairs"""
    E2E = json.load(open(os.path.join(_enc.ROOT, "tests", "golden", "e2e_v1.json")))
    w, h, n, seed, cli = _e2e.PLACEBO_CASES[name]
    threads = int(cli[cli.index("-m") + 1]) if "-m" in cli else 1
    cli = [a for i, a in enumerate(cli) if a != "-m" and (i == 0 or cli[i - 1] != "-m")]
    out = _enc.encode_cpu(_enc.config(w, h, cli, threads), [_clip(str(tmp_path_factory.getbasetemp()), name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.PLACEBO_BATCH_CASES))
def test_walk_decides_every_ctu_of_a_preset_placebo_batch(name, walk_engine, tmp_path_factory):
    w, h, gops, frames, seed, cli, threads = _enc.PLACEBO_BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _clip(str(tmp_path_factory.getbasetemp()), name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


def test_walk_decides_every_ctu_of_a_preset_slow_batch(walk_engine, tmp_path_factory):
    name = "slow_gops_192x128_moving_m3"
    w, h, gops, frames, seed, cli, threads = _enc.SLOW_BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _clip(str(tmp_path_factory.getbasetemp()), name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]
