"""The mode decision of I-picture CTUs on the GPU (xeve_hip_mode_analyze_ctu_intra_jobs) against the pinned oracle: whole small pictures coded CTU by CTU, the
pictures of a case as the chains of one call.  After every CTU: the CTU's data (split modes, prediction modes, depths, nnz, map fields, levels, reconstruction)
byte for byte, the coder state handed to the next CTU, the cost as the bit pattern of the double; at the end the reconstructed pictures and the 4x4-unit maps."""
import numpy as np
import pytest

from _libs import SBAC_DTYPE
from _tree_cases import CASES, CTU_DATA_DTYPE, CTU_JOB_DTYPE, INTER_CASES, make_case, make_inter_case, run_oracle_inter_picture, run_oracle_picture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _every_test_here_on_each_walk(each_walk):  # (tests/conftest.py: the library's choice, the composed walk pinned with and without its side stream, the fused kernel with 3 chains per team)
    return each_walk


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


# ---- P / B slices ---------------------------------------------------------------------------------------------------------------------------------------------
def run_hip_inter_case(c):
    import ctypes as C

    import torch
    import xeve_amd
    from test_hip_inter import hip_params
    from xeve_amd import device as D
    from xeve_amd import lib
    from _mc_cases import refpic_table

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    refs, org = c["refs"], c["org"]
    dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
    lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off)
    dorg = [torch.from_numpy(x).to(dev) for x in org]
    org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in c["mod"]]
    m = c["maps"]
    ms, mc = (torch.from_numpy(m[k].view(np.int32).copy()).to(dev) for k in ("scu", "cu_mode"))
    mi, mt, mv, mr = (torch.from_numpy(m[k].copy()).to(dev) for k in ("ipm", "tidx", "mv", "refi"))
    col = [torch.from_numpy(a.copy()).to(dev) for a in c["col"]]
    P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
    I = lib.TreeInter()
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = dev_tab.ctypes.data, refs["s_l"], refs["s_c"], hip_params(c["ipar"])
    I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = mv.data_ptr(), mr.data_ptr(), col[0].data_ptr(), col[1].data_ptr(), c["ecu_depth"]
    states = torch.from_numpy(c["entry"][0:1].view(np.uint8).copy()).to(dev)
    per_ctu = []
    for (x, y) in c["order"]:
        jobs = np.zeros(1, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"] = x, y
        out, nxt, cost = D.mode_analyze_ctu_jobs(org_ptrs, refs["s_l"], refs["s_c"], [t.data_ptr() for t in mod], mod[0].shape[1], mod[1].shape[1], ms, mi, mt, mc, states, P,
                                                 torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), inter=I)
        torch.cuda.synchronize()
        per_ctu.append((out.cpu().numpy().reshape(-1).view(CTU_DATA_DTYPE), nxt.cpu().numpy().reshape(-1).view(SBAC_DTYPE), cost.cpu().numpy()))
        states = nxt.clone()
    final = dict(mod=[t.cpu().numpy() for t in mod], scu=ms.cpu().numpy().view(np.uint32), ipm=mi.cpu().numpy(), cu_mode=mc.cpu().numpy().view(np.uint32), mv=mv.cpu().numpy(),
                 refi=mr.cpu().numpy())
    return per_ctu, final


@pytest.mark.parametrize("case", INTER_CASES, ids=[str(c[0]) for c in INTER_CASES])
def test_hip_ctu_mode_decision_of_p_and_b_slices_matches_oracle(case):
    """mode_coding_unit on the device: the whole inter analysis, the intra analysis cut against the inter winner where that has a residual, the cheaper one kept;
    the motion maps updated CU by CU (they are the next CU's merge / MVP candidates); a skipped CU ends the split from a POC-dependent depth on"""
    c = make_inter_case(*case)
    got, final = run_hip_inter_case(c)
    exp = run_oracle_inter_picture(c)  # updates c["mod"], c["maps"] in place
    modes = set()
    for k in range(len(c["order"])):
        d, nb, cost = got[k]
        ed, enb, ecost = exp[k]
        for f in CTU_DATA_DTYPE.names:
            assert np.array_equal(d[f][0], ed[f][0]), (case, "ctu", k, f, np.argwhere(d[f][0] != ed[f][0])[:4].tolist())
        assert nb[0:1].tobytes() == enb.tobytes(), (case, k, "coder state")
        assert np.float64(cost[0]).tobytes() == np.float64(ecost).tobytes(), (case, k, cost[0], ecost)
        modes |= set(np.unique(ed["pred_mode"][0]).tolist())
    for j in range(3 if c["idc"] else 1):
        assert np.array_equal(final["mod"][j], c["mod"][j]), (case, "picture", j)
    for f in ("scu", "ipm", "cu_mode", "mv", "refi"):
        assert np.array_equal(final[f].reshape(c["maps"][f].shape), c["maps"][f]), (case, "map", f)
    assert len(modes) >= 3, modes  # intra, inter and skip CUs all occur


def test_hip_ctu_rows_of_a_b_picture_as_chains_of_one_call():
    """two CTU rows of one B picture walked as a wavefront: row 1 starts when row 0 is two CTUs ahead (its up-right neighbour is decided), and from then on each call
    advances BOTH rows -- two chains of one picture in lockstep, sharing the maps and the picture being reconstructed.  Each row carries its own coder state.
    The oracle walks the same CTUs one at a time in the same order."""
    import ctypes as C

    import torch
    import xeve_amd
    from test_hip_inter import hip_params
    from xeve_amd import device as D
    from xeve_amd import lib
    from _mc_cases import refpic_table
    from _tree_cases import TreeInter, oracle_tree_any
    from _libs import c_void_p, ptr
    from _sbac_cases import make_states

    c = make_inter_case(4201, 256, 128, 10, 1, 0, 1, 0, 0.0)
    steps = [[(0, 0)], [(64, 0)], [(128, 0), (0, 64)], [(192, 0), (64, 64)], [(128, 64)], [(192, 64)]]
    entry = make_states(np.random.default_rng(5), 2)  # one coder state per row
    # device
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    refs, org = c["refs"], c["org"]
    dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
    lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off)
    dorg = [torch.from_numpy(x).to(dev) for x in org]
    org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in c["mod"]]
    m = c["maps"]
    ms, mc = (torch.from_numpy(m[k].view(np.int32).copy()).to(dev) for k in ("scu", "cu_mode"))
    mi, mt, mv, mr = (torch.from_numpy(m[k].copy()).to(dev) for k in ("ipm", "tidx", "mv", "refi"))
    col = [torch.from_numpy(a.copy()).to(dev) for a in c["col"]]
    P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
    I = lib.TreeInter()
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = dev_tab.ctypes.data, refs["s_l"], refs["s_c"], hip_params(c["ipar"])
    I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = mv.data_ptr(), mr.data_ptr(), col[0].data_ptr(), col[1].data_ptr(), c["ecu_depth"]
    states = torch.from_numpy(entry.view(np.uint8).copy()).to(dev)
    got = {}
    for ctus in steps:
        jobs = np.zeros(len(ctus), CTU_JOB_DTYPE)
        for k, (x, y) in enumerate(ctus):
            jobs["x"][k], jobs["y"][k], jobs["sbac"][k] = x, y, y // 64
        out, nxt, cost = D.mode_analyze_ctu_jobs(org_ptrs, refs["s_l"], refs["s_c"], [t.data_ptr() for t in mod], mod[0].shape[1], mod[1].shape[1], ms, mi, mt, mc, states, P,
                                                 torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), inter=I)
        torch.cuda.synchronize()
        o, nb, cs = out.cpu().numpy().reshape(-1).view(CTU_DATA_DTYPE), nxt.cpu().numpy().reshape(-1).view(SBAC_DTYPE), cost.cpu().numpy()
        st = states.cpu().numpy().reshape(-1).view(SBAC_DTYPE).copy()
        for k, (x, y) in enumerate(ctus):
            got[(x, y)] = (o[k:k + 1].copy(), nb[k:k + 1].copy(), cs[k])
            st[y // 64] = nb[k]
        states = torch.from_numpy(st.view(np.uint8).copy()).to(dev)
    # oracle, CTU by CTU in the same order
    O = oracle_tree_any()
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    TI = TreeInter()
    TI.refp, TI.s_ref_l, TI.s_ref_c, TI.ipar = tab.ctypes.data, refs["s_l"], refs["s_c"], c["ipar"]
    TI.map_mv, TI.map_refi, TI.col0, TI.col1, TI.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, c["col"][0].ctypes.data, c["col"][1].ctypes.data, c["ecu_depth"]
    orgp = (c_void_p * 3)(int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"])
    modp = (c_void_p * 3)(*[a.ctypes.data for a in c["mod"]])
    st = entry.copy()
    for ctus in steps:
        for (x, y) in ctus:
            d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
            cost = O.xo_mode_analyze_ctu(orgp, refs["s_l"], refs["s_c"], modp, c["mod"][0].shape[1], c["mod"][1].shape[1], ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]),
                                         ptr(m["cu_mode"]), ptr(st[y // 64:y // 64 + 1]), C.byref(c["P"]), C.byref(TI), x, y, ptr(d), ptr(nb))
            st[y // 64] = nb[0]
            gd, gnb, gcost = got[(x, y)]
            for f in CTU_DATA_DTYPE.names:
                assert np.array_equal(gd[f][0], d[f][0]), ((x, y), f)
            assert gnb.tobytes() == nb.tobytes() and np.float64(gcost).tobytes() == np.float64(cost).tobytes(), (x, y)
    assert np.array_equal(mod[0].cpu().numpy(), c["mod"][0]) and np.array_equal(mv.cpu().numpy(), m["mv"]) and np.array_equal(ms.cpu().numpy().view(np.uint32), m["scu"])


def test_hip_ctu_mode_decision_with_the_lane_serial_node_kernel(tmp_path, each_walk):
    """XEVE_HIP_TREE_LANE=1 (off by default: measured slower): the 4x4 / 8x8 nodes of the COMPOSED walk decided by one lane per chain (csrc/cu_lane.h).  The
    switch is read once per process, so a fresh interpreter runs three of the cases above with it on (composed walk pinned); the same comparison against the oracle must hold."""
    import os
    import subprocess
    import sys

    if each_walk != "composed":
        pytest.skip("the lane-serial node kernel belongs to the composed walk: run once, with it pinned")
    env = dict(os.environ, XEVE_HIP_TREE_LANE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_hip_tree.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "matches_oracle and composed and (3101 or 3104 or 4103)"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "9 passed" in p.stdout, p.stdout[-1500:] + p.stderr[-500:]  # (three cases x {side stream, one stream, a side stream per level})


def test_hip_ctu_host_form_over_two_b_pictures():
    """xeve_hip_mode_analyze_ctu_host (what ctx->fn_mode_analyze_lcu is pointed at) called directly, CTU by CTU, over TWO B pictures of one size one after the other:
    host planes and maps, resident pictures announced per picture, the per-thread device buffers reused -- what the first picture left in them (every unit coded)
    must not leak into the second: the left neighbours of the CUs in a CTU's bottom rows reach into the CTU row below, which is not coded yet."""
    import ctypes as C

    import xeve_amd
    from test_hip_inter import hip_params
    from xeve_amd import device as D
    from xeve_amd import lib
    from _mc_cases import PAD_L, refpic_table

    xeve_amd.init(0)
    L = lib.load()
    for seed in (4301, 4302):  # (this pair failed before the units below the CTU were handed over: picture 1 had left them "coded" in the device buffers)
        c = make_inter_case(seed, 192, 192, 10, 1, 0, 1, 0, 0.0)
        refs, org = c["refs"], c["org"]
        exp_c = make_inter_case(seed, 192, 192, 10, 1, 0, 1, 0, 0.0)
        exp = run_oracle_inter_picture(exp_c)
        lib.check(L.xeve_hip_picture_begin())
        tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
        P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
        I = lib.TreeInter()
        m = c["maps"]
        I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, refs["s_l"], refs["s_c"], hip_params(c["ipar"])
        I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, c["col"][0].ctypes.data, c["col"][1].ctypes.data, c["ecu_depth"]
        I.coef_l, I.coef_c = D.baseline_coef_l().ctypes.data, D.baseline_coef_c().ctypes.data
        orgp = (C.c_void_p * 3)(int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"])
        modp = (C.c_void_p * 3)(*[a.ctypes.data for a in c["mod"]])
        state = c["entry"][0:1].copy()
        for k, (x, y) in enumerate(c["order"]):
            d, nb, cost = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE), np.zeros(1, np.float64)
            lib.check(L.xeve_hip_mode_analyze_ctu_host(orgp, refs["s_l"], refs["s_c"], modp, c["mod"][0].shape[1], c["mod"][1].shape[1], m["scu"].ctypes.data, m["ipm"].ctypes.data,
                                                       m["tidx"].ctypes.data, m["cu_mode"].ctypes.data, state.ctypes.data, C.byref(P), C.byref(I), PAD_L, PAD_L, x, y,
                                                       d.ctypes.data, nb.ctypes.data, cost.ctypes.data))
            ed, enb, ecost = exp[k]
            for f in CTU_DATA_DTYPE.names:
                assert np.array_equal(d[f][0], ed[f][0]), (seed, "ctu", k, f)
            assert nb.tobytes() == enb.tobytes() and cost.tobytes() == np.float64(ecost).tobytes(), (seed, k)
            state = nb.copy()
        for j in range(3):
            assert np.array_equal(c["mod"][j], exp_c["mod"][j]), (seed, "picture", j)
        for f in ("scu", "ipm", "cu_mode", "mv", "refi"):
            assert np.array_equal(m[f], exp_c["maps"][f]), (seed, "map", f)


@pytest.mark.parametrize("case", [CASES[0], CASES[3]], ids=["3101", "3104"])
def test_hip_i_pictures_decided_and_written_ctu_by_ctu_on_the_device(case):
    """the closed loop of a chain on the device: every CTU is decided (xeve_hip_mode_analyze_ctu_jobs) from the WRITER's coder state and then written
    (xeve_hip_eco_ctu_jobs: xeve_eco_tree, one lane per chain), which advances that state in place -- the next CTU enters with it (xeve_enc.c:139), no host in between.
    The pictures of the case are the chains.  Against the oracle's same chain (xo_mode_analyze_ctu_intra + xo_eco_ctu, both pinned beside the live encoder): the writer's
    state after every CTU, the bytes that came out, at the end the maps and the reconstructed pictures."""
    import ctypes as C

    import torch
    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib
    from _libs import c_int, c_void_p, oracle, ptr
    from _tree_cases import oracle_tree

    c = make_case(*case)
    n = c["npic"]
    entry = c["entry"].copy()
    entry["bitcounter"] = 0
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    org = [torch.from_numpy(a.copy()).to(dev) for a in c["org"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in c["mod"]]
    m = c["maps"]
    ms, mi, mt, mc = (torch.from_numpy(m[k].view(np.int32 if m[k].dtype == np.uint32 else m[k].dtype).copy()).to(dev) for k in ("scu", "ipm", "tidx", "cu_mode"))
    P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
    EP = lib.EcoParams()
    EP.chroma_format_idc, EP.slice_type, EP.log2_ctu, EP.pic_w, EP.pic_h, EP.w_scu, EP.h_scu = c["idc"], 2, P.log2_ctu, P.pic_w, P.pic_h, P.ip.w_scu, P.ip.h_scu
    states = torch.from_numpy(entry.view(np.uint8).copy()).to(dev)
    pe = (org[0][0].numel(), org[1][0].numel(), mod[0][0].numel(), mod[1][0].numel(), m["scu"].shape[1])
    got = []
    for (x, y) in c["order"]:
        jobs = np.zeros(n, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"], jobs["sbac"], jobs["pic"] = x, y, np.arange(n), np.arange(n)
        jt = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        out, _, _ = D.mode_analyze_ctu_jobs([t.data_ptr() for t in org], org[0].shape[2], org[1].shape[2], [t.data_ptr() for t in mod], mod[0].shape[2], mod[1].shape[2], ms, mi, mt,
                                            mc, states, P, jt, pic_elems=pe)
        by, nb = D.eco_ctu_jobs(out, states, EP, ms, mi, mt, mc, jt, map_pic_elems=m["scu"].shape[1])
        torch.cuda.synchronize()
        got.append((states.cpu().numpy().reshape(-1).view(SBAC_DTYPE).copy(), by.cpu().numpy(), nb.cpu().numpy()))
    eby, enb = D.eco_tile_end_jobs(states, jt)  # the tile's end: with it the bytes of a chain are the picture's slice data
    torch.cuda.synchronize()
    eby, enb = eby.cpu().numpy(), enb.cpu().numpy()
    # the oracle's chain, picture by picture
    O = oracle_tree()
    OE = oracle()
    OE.xo_eco_ctu.restype = c_int
    OE.xo_eco_ctu.argtypes = [c_void_p, c_void_p, C.c_void_p, c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_void_p, c_int]
    total = 0
    for p in range(n):
        orgp = (c_void_p * 3)(*[a[p].ctypes.data for a in c["org"]])
        modp = (c_void_p * 3)(*[a[p].ctypes.data for a in c["mod"]])
        state = entry[p:p + 1].copy()
        for k, (x, y) in enumerate(c["order"]):
            d, nx = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_mode_analyze_ctu_intra(orgp, c["org"][0].shape[2], c["org"][1].shape[2], modp, c["mod"][0].shape[2], c["mod"][1].shape[2], ptr(m["scu"][p]), ptr(m["ipm"][p]),
                                        ptr(m["tidx"][p]), ptr(m["cu_mode"][p]), ptr(state), C.byref(c["P"]), x, y, ptr(d), ptr(nx))
            ctu, w_scu = 1 << c["P"].log2_ctu, c["P"].ip.w_scu
            for j in range(min(ctu, c["h"] - y) // 4):  # mode_analyze_lcu's tail: the CTU's coded flags reset
                g = (y // 4 + j) * w_scu + x // 4
                m["scu"][p][g:g + min(ctu, c["w"] - x) // 4] &= np.uint32(0x7FFFFFFF)
            eb = np.zeros(1 << 15, np.uint8)
            ne = OE.xo_eco_ctu(ptr(state), ptr(d), C.addressof(c["P"]), ptr(np.zeros(2, np.int32)), ptr(m["scu"][p]), ptr(m["ipm"][p]), ptr(m["tidx"][p]), ptr(m["cu_mode"][p]), x, y,
                               ptr(eb), eb.size)
            gs, gb, gn = got[k]
            for f in ("range", "code", "code_bits", "stacked_ff", "stacked_zero", "pending_byte", "is_pending_byte", "bin_counter", "ctx"):
                assert np.array_equal(gs[f][p], state[f][0]), (case, "picture", p, "ctu", k, f)
            assert int(gn[p]) == ne and np.array_equal(gb[p][:ne], eb[:ne]), (case, p, k, "bytes", int(gn[p]), ne)
            total += ne
        OE.xo_eco_tile_end.restype = c_int
        OE.xo_eco_tile_end.argtypes = [c_void_p, c_void_p, c_int]
        tb = np.zeros(64, np.uint8)
        nt = OE.xo_eco_tile_end(ptr(state), ptr(tb), 64)
        assert int(enb[p]) == nt and np.array_equal(eby[p][:nt], tb[:nt]), (case, p, "tile end", int(enb[p]), nt)
    assert total > 100
    for j in range(3 if c["idc"] else 1):
        assert np.array_equal(mod[j].cpu().numpy(), c["mod"][j]), (case, "picture", j)
    assert np.array_equal(ms.cpu().numpy().view(np.uint32).reshape(m["scu"].shape), m["scu"]) and np.array_equal(mc.cpu().numpy().view(np.uint32).reshape(m["cu_mode"].shape), m["cu_mode"])


@pytest.mark.parametrize("case", [INTER_CASES[1], INTER_CASES[2]], ids=["4102", "4103"])
def test_hip_b_picture_decided_and_written_ctu_by_ctu_on_the_device(case):
    """the same closed loop for a B picture (one chain: its CTUs in raster order): decided from the writer's state, written -- skip, direct, uni- and bi-predicted and intra
    CUs in the writer's syntax -- the state advanced in place, the tile's end; against the oracle's chain"""
    import ctypes as C

    import torch
    import xeve_amd
    from test_hip_inter import hip_params
    from xeve_amd import device as D
    from xeve_amd import lib
    from _libs import c_int, c_void_p, oracle, ptr
    from _mc_cases import refpic_table
    from _tree_cases import TreeInter, oracle_tree_any

    c = make_inter_case(*case)
    exp_c = make_inter_case(*case)
    entry = c["entry"][0:1].copy()
    entry["bitcounter"] = 0
    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    refs, org = c["refs"], c["org"]
    dplanes = [[torch.from_numpy(x).to(dev) for x in pic] for pic in refs["pics"]]
    lut = {id(x): t for pic, dp in zip(refs["pics"], dplanes) for x, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off)
    dorg = [torch.from_numpy(x).to(dev) for x in org]
    org_ptrs = [dorg[0].data_ptr() + 2 * refs["org_l"], dorg[1].data_ptr() + 2 * refs["org_c"], dorg[2].data_ptr() + 2 * refs["org_c"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in c["mod"]]
    m = c["maps"]
    ms, mc = (torch.from_numpy(m[k].view(np.int32).copy()).to(dev) for k in ("scu", "cu_mode"))
    mi, mt, mv, mr = (torch.from_numpy(m[k].copy()).to(dev) for k in ("ipm", "tidx", "mv", "refi"))
    col = [torch.from_numpy(a.copy()).to(dev) for a in c["col"]]
    P = lib.TreeParams.from_buffer_copy(bytes(c["P"]))
    I = lib.TreeInter()
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = dev_tab.ctypes.data, refs["s_l"], refs["s_c"], hip_params(c["ipar"])
    I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = mv.data_ptr(), mr.data_ptr(), col[0].data_ptr(), col[1].data_ptr(), c["ecu_depth"]
    nref = (c["ipar"].rdo.num_refp[0], c["ipar"].rdo.num_refp[1])
    EP = lib.EcoParams()
    EP.chroma_format_idc, EP.slice_type, EP.log2_ctu, EP.pic_w, EP.pic_h, EP.w_scu, EP.h_scu = c["idc"], c["slice_type"], 6, P.pic_w, P.pic_h, P.ip.w_scu, P.ip.h_scu
    EP.num_refp[0], EP.num_refp[1] = nref
    states = torch.from_numpy(entry.view(np.uint8).copy()).to(dev)
    got = []
    for (x, y) in c["order"]:
        jobs = np.zeros(1, CTU_JOB_DTYPE)
        jobs["x"], jobs["y"] = x, y
        jt = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        out, _, _ = D.mode_analyze_ctu_jobs(org_ptrs, refs["s_l"], refs["s_c"], [t.data_ptr() for t in mod], mod[0].shape[1], mod[1].shape[1], ms, mi, mt, mc, states, P, jt, inter=I)
        by, nb = D.eco_ctu_jobs(out, states, EP, ms, mi, mt, mc, jt)
        torch.cuda.synchronize()
        got.append((states.cpu().numpy().reshape(-1).view(SBAC_DTYPE).copy(), by.cpu().numpy()[0], int(nb.cpu().numpy()[0])))
    eby, enb = D.eco_tile_end_jobs(states, jt)
    torch.cuda.synchronize()
    # the oracle's chain
    O, OE = oracle_tree_any(), oracle()
    OE.xo_eco_ctu.restype = c_int
    OE.xo_eco_ctu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_void_p, c_int]
    OE.xo_eco_tile_end.restype = c_int
    OE.xo_eco_tile_end.argtypes = [c_void_p, c_void_p, c_int]
    erefs, eorg, em = exp_c["refs"], exp_c["org"], exp_c["maps"]
    tab = refpic_table(erefs, lambda a, off: int(a.ctypes.data) + 2 * off)
    TI = TreeInter()
    TI.refp, TI.s_ref_l, TI.s_ref_c, TI.ipar = tab.ctypes.data, erefs["s_l"], erefs["s_c"], exp_c["ipar"]
    TI.map_mv, TI.map_refi, TI.col0, TI.col1, TI.ecu_depth = em["mv"].ctypes.data, em["refi"].ctypes.data, exp_c["col"][0].ctypes.data, exp_c["col"][1].ctypes.data, exp_c["ecu_depth"]
    orgp = (c_void_p * 3)(int(eorg[0].ctypes.data) + 2 * erefs["org_l"], int(eorg[1].ctypes.data) + 2 * erefs["org_c"], int(eorg[2].ctypes.data) + 2 * erefs["org_c"])
    modp = (c_void_p * 3)(*[a.ctypes.data for a in exp_c["mod"]])
    state, nr, total = entry.copy(), np.array(nref, np.int32), 0
    for k, (x, y) in enumerate(c["order"]):
        d, nx = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
        O.xo_mode_analyze_ctu(orgp, erefs["s_l"], erefs["s_c"], modp, exp_c["mod"][0].shape[1], exp_c["mod"][1].shape[1], ptr(em["scu"]), ptr(em["ipm"]), ptr(em["tidx"]),
                              ptr(em["cu_mode"]), ptr(state), C.byref(exp_c["P"]), C.byref(TI), x, y, ptr(d), ptr(nx))
        for j in range(min(64, c["h"] - y) // 4):
            g = (y // 4 + j) * (c["w"] // 4) + x // 4
            em["scu"][g:g + min(64, c["w"] - x) // 4] &= np.uint32(0x7FFFFFFF)
        eb = np.zeros(1 << 15, np.uint8)
        ne = OE.xo_eco_ctu(ptr(state), ptr(d), C.addressof(exp_c["P"]), ptr(nr), ptr(em["scu"]), ptr(em["ipm"]), ptr(em["tidx"]), ptr(em["cu_mode"]), x, y, ptr(eb), eb.size)
        gs, gb, gn = got[k]
        for f in ("range", "code", "code_bits", "stacked_ff", "stacked_zero", "pending_byte", "is_pending_byte", "bin_counter", "ctx"):
            assert np.array_equal(gs[f][0], state[f][0]), (case, "ctu", k, f)
        assert gn == ne and np.array_equal(gb[:ne], eb[:ne]), (case, k, "bytes", gn, ne)
        total += ne
    tb = np.zeros(64, np.uint8)
    nt = OE.xo_eco_tile_end(ptr(state), ptr(tb), 64)
    assert int(enb.cpu().numpy()[0]) == nt and np.array_equal(eby.cpu().numpy()[0][:nt], tb[:nt]), (case, "tile end")
    assert total > 10
    assert np.array_equal(ms.cpu().numpy().view(np.uint32), em["scu"]) and np.array_equal(mv.cpu().numpy(), em["mv"]) and np.array_equal(mod[0].cpu().numpy(), exp_c["mod"][0])

