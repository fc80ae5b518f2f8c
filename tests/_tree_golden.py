"""tests/golden/tree_v1.npz (made by tests/golden/make_tree_golden.py from the reference encoder): CTUs of real encodes -- what the CTU mode decision was handed and what
the REFERENCE made of it.  The loader turns a record into the arrays the oracle / the library take; `same_as_reference` compares a walk's products with the
reference's the way oracle/ref_shadow.c's shadow mode does (fields the reference leaves stale are not compared: motion data of unused lists, levels behind nnz = 0)."""
import ctypes as C
import os

import numpy as np

from _libs import SBAC_DTYPE, InterParams, REFPIC_DTYPE, c_int, c_void_p, ptr
from _tree_cases import CTU_DATA_DTYPE, TreeInter, TreeParams, oracle_tree, oracle_tree_any

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree_v1.npz")


def load():
    g = np.load(GOLDEN)
    for k in range(int(g["n_records"][0])):
        f = lambda name: g["r%d_%s" % (k, name)]
        has = lambda name: ("r%d_%s" % (k, name)) in g
        head = f("head").view(np.int32)
        P = TreeParams.from_buffer_copy(f("params").tobytes())
        w, h, idc = P.pic_w, P.pic_h, P.ip.chroma_format_idc
        ws, hs = (1 if idc in (1, 2) else 0), (1 if idc == 1 else 0)
        wc, hc = (w >> ws, h >> hs) if idc else (1, 1)
        pl = lambda name, hh, ww: f(name).view(np.int16).reshape(hh, ww).copy() if has(name) else np.zeros((1, 1), np.int16)
        r = dict(k=k, clip=f("clip").tobytes().decode(), poc=int(head[0]), slice_type=int(head[1]), x0=int(head[2]), y0=int(head[3]), P=P, w=w, h=h, idc=idc, ws=ws, hs=hs,
                 entry=f("entry").view(SBAC_DTYPE).copy(), org=[pl("org_y", h, w), pl("org_u", hc, wc), pl("org_v", hc, wc)],
                 mod=[pl("mod_y", h, w), pl("mod_u", hc, wc), pl("mod_v", hc, wc)],
                 maps=dict(scu=f("map_scu").view(np.uint32).copy(), ipm=f("map_ipm").view(np.int8).copy(), tidx=f("map_tidx").copy(), cu_mode=f("map_cu_mode").view(np.uint32).copy()),
                 ref_ctu=f("ref_ctu").view(CTU_DATA_DTYPE).copy(), ref_next=f("ref_next").view(SBAC_DTYPE).copy(),
                 ref_maps=dict(scu=f("ref_scu").view(np.uint32), ipm=f("ref_ipm").view(np.int8), cu_mode=f("ref_cu_mode").view(np.uint32)),
                 ref_mod=[pl("ref_mod_y", h, w), pl("ref_mod_u", hc, wc), pl("ref_mod_v", hc, wc)])
        if has("wr_head"):  # the reference WRITER's side of this CTU (xeve_eco_tree): the coder state it left, its bytes, the unit flags after
            wh = f("wr_head").view(np.int32)
            r["wr"] = dict(num_refp=(int(wh[5]), int(wh[6])), state=f("wr_state").view(SBAC_DTYPE).copy(), bytes=f("wr_bytes").copy() if int(wh[7]) else np.zeros(0, np.uint8),
                           scu=f("wr_scu").view(np.uint32), cu_mode=f("wr_cu_mode").view(np.uint32))
        if r["slice_type"] != 2:
            rh = f("ref_head").view(np.int32)
            nr, pad_l, pad_c, s_l, s_c = (int(rh[0]), int(rh[1])), int(rh[2]), int(rh[3]), int(rh[4]), int(rh[5])
            nscu = (w // 4) * (h // 4)
            r["maps"]["mv"], r["maps"]["refi"] = f("map_mv").view(np.int16).reshape(nscu, 2, 2).copy(), f("map_refi").view(np.int8).reshape(nscu, 2).copy()
            r["ref_maps"]["mv"], r["ref_maps"]["refi"] = f("ref_mv").view(np.int16).reshape(nscu, 2, 2), f("ref_refi").view(np.int8).reshape(nscu, 2)
            r["col"] = [f("col0").view(np.int16).reshape(nscu, 2, 2).copy(), f("col1").view(np.int16).reshape(nscu, 2, 2).copy()]
            r["ipar"] = InterParams.from_buffer_copy(f("inter_params").tobytes())
            r["ecu_depth"], r["nr"], r["pad"], r["s_ref"] = int(head[7]), nr, (pad_l, pad_c), (s_l, s_c)
            planes, pocs = {}, {}
            for l in range(2):
                for i in range(nr[l]):
                    key = lambda c: g["plane_" + f("ref%d_%d_%s" % (i, l, c)).tobytes().decode()]
                    planes[(i, l)] = [key("y"), key("u") if idc else None, key("v") if idc else None]
                    pocs[(i, l)] = int(f("ref%d_%d_poc" % (i, l)).view(np.int32)[0])
            r["ref_planes"], r["ref_pocs"] = planes, pocs
        yield r


def refpic_table_of(r, addr_of):
    """REFPIC_DTYPE array [refi * 2 + list] over the record's padded reference planes; addr_of(plane_array, element_offset) -> address (host or device)"""
    n = max(r["nr"])
    t = np.zeros(2 * max(n, 1), REFPIC_DTYPE)
    (pad_l, pad_c), (s_l, s_c) = r["pad"], r["s_ref"]
    for (i, l), pl in r["ref_planes"].items():
        t["y"][2 * i + l] = addr_of(pl[0], pad_l * s_l + pad_l)
        if r["idc"]:
            t["u"][2 * i + l], t["v"][2 * i + l] = addr_of(pl[1], pad_c * s_c + pad_c), addr_of(pl[2], pad_c * s_c + pad_c)
        t["poc"][2 * i + l] = r["ref_pocs"][(i, l)]
    if r["slice_type"] == 1:
        t[1] = t[0]  # (P slices never read list 1; keep the table addressable)
    return t


def run_oracle(r):
    """the oracle's walk of the record's CTU on copies of its inputs -> (ctu data, next state, maps after, planes after)"""
    mod = [a.copy() for a in r["mod"]]
    m = {k: v.copy() for k, v in r["maps"].items()}
    d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
    org = (c_void_p * 3)(*[a.ctypes.data for a in r["org"]])
    modp = (c_void_p * 3)(*[a.ctypes.data for a in mod])
    if r["slice_type"] == 2:
        oracle_tree().xo_mode_analyze_ctu_intra(org, r["org"][0].shape[1], r["org"][1].shape[1], modp, mod[0].shape[1], mod[1].shape[1], ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]),
                                                ptr(m["cu_mode"]), ptr(r["entry"]), C.byref(r["P"]), r["x0"], r["y0"], ptr(d), ptr(nb))
    else:
        tab = refpic_table_of(r, lambda a, off: int(a.ctypes.data) + 2 * off)
        I = TreeInter()
        I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, r["s_ref"][0], r["s_ref"][1], r["ipar"]
        I.map_mv, I.map_refi, I.col0, I.col1, I.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, r["col"][0].ctypes.data, r["col"][1].ctypes.data, r["ecu_depth"]
        oracle_tree_any().xo_mode_analyze_ctu(org, r["org"][0].shape[1], r["org"][1].shape[1], modp, mod[0].shape[1], mod[1].shape[1], ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]),
                                              ptr(m["cu_mode"]), ptr(r["entry"]), C.byref(r["P"]), C.byref(I), r["x0"], r["y0"], ptr(d), ptr(nb))
    return d, nb, m, mod


def same_as_reference(r, d, nb, m, mod):
    """(ctu data, next state, maps after, planes after) of a walk against what the reference left behind for the record's CTU"""
    e, what = r["ref_ctu"], ("record", r["k"], r["clip"], "poc", r["poc"], "ctu at", r["x0"], r["y0"])
    w_scu, x0, y0, ctu, idc, inter = r["w"] // 4, r["x0"], r["y0"], 1 << r["P"].log2_ctu, r["idc"], r["slice_type"] != 2
    nu = ctu // 4
    wu, hu = min(nu, (r["w"] - x0) // 4), min(nu, (r["h"] - y0) // 4)
    assert np.array_equal(d["split_mode"][0], e["split_mode"][0]), (what, "split_mode")
    units = np.array([j * nu + i for j in range(hu) for i in range(wu)])
    glob = np.array([(y0 // 4 + j) * w_scu + x0 // 4 + i for j in range(hu) for i in range(wu)])
    for f in ("pred_mode", "depth", "map_scu", "map_cu_mode"):
        assert np.array_equal(d[f][0][units], e[f][0][units]), (what, f)
    for c in range(3 if idc else 1):
        assert np.array_equal(d["nnz"][0][c][units], e["nnz"][0][c][units]), (what, "nnz", c)
    for c in range(2 if idc else 1):
        assert np.array_equal(d["ipm"][0][c][units], e["ipm"][0][c][units]), (what, "ipm", c)
    if inter:
        pm = e["pred_mode"][0][units]
        assert np.array_equal(d["refi"][0][units], e["refi"][0][units]) and np.array_equal(d["mv"][0][units], e["mv"][0][units]), (what, "motion")
        for l in range(2):
            used = (pm != 0) & (e["refi"][0][units, l] >= 0) & (pm != 3)
            assert np.array_equal(d["mvp_idx"][0][units, l][used], e["mvp_idx"][0][units, l][used]), (what, "mvp_idx", l)
            coded = used & (pm == 1)
            assert np.array_equal(d["mvd"][0][units, l][coded], e["mvd"][0][units, l][coded]), (what, "mvd", l)
    for c in range(3 if idc else 1):
        sx, sy = (r["ws"], r["hs"]) if c else (0, 0)
        cs, ww, hh = ctu >> sx, (wu * 4) >> sx, (hu * 4) >> sy
        got_c, exp_c = d["coef"][0][c].reshape(-1, cs)[:hh, :ww], e["coef"][0][c].reshape(-1, cs)[:hh, :ww]
        yy, xx = np.mgrid[0:hh, 0:ww]
        coded = e["nnz"][0][c][((yy << sy) >> 2) * nu + ((xx << sx) >> 2)] != 0  # (a CU without coded levels keeps stale ones in the reference)
        assert np.array_equal(got_c[coded], exp_c[coded]), (what, "coef", c)
        assert np.array_equal(d["reco"][0][c].reshape(-1, cs)[:hh, :ww], e["reco"][0][c].reshape(-1, cs)[:hh, :ww]), (what, "reco", c)
        px, py = x0 >> sx, y0 >> sy
        assert np.array_equal(mod[c][py:py + hh, px:px + ww], r["ref_mod"][c][py:py + hh, px:px + ww]), (what, "picture", c)
    assert nb.tobytes() == r["ref_next"].tobytes(), (what, "exit coder state")
    assert np.array_equal(m["scu"][glob], r["ref_maps"]["scu"][glob] | np.uint32(1 << 31)), (what, "map_scu")  # (the reference has reset the coded flags on return)
    assert np.array_equal(m["ipm"][glob], r["ref_maps"]["ipm"][glob]) and np.array_equal(m["cu_mode"][glob], r["ref_maps"]["cu_mode"][glob]), (what, "maps")
    if inter:
        assert np.array_equal(m["mv"][glob], r["ref_maps"]["mv"][glob]) and np.array_equal(m["refi"][glob], r["ref_maps"]["refi"][glob]), (what, "motion maps")


def run_walk(r, dev, call):
    """the record's CTU through `call` (xeve_amd.device.mode_analyze_ctu_jobs, or a stand-in with its signature) with every operand a torch tensor on `dev`
    -> (ctu data, next state, maps after, planes after) as numpy, like run_oracle"""
    import torch
    from test_hip_inter import hip_params
    from xeve_amd import lib

    org = [torch.from_numpy(a.copy()).to(dev) for a in r["org"]]
    mod = [torch.from_numpy(a.copy()).to(dev) for a in r["mod"]]
    m = r["maps"]
    ms, mc = (torch.from_numpy(m[k].view(np.int32).copy()).to(dev) for k in ("scu", "cu_mode"))
    mi, mt = (torch.from_numpy(m[k].copy()).to(dev) for k in ("ipm", "tidx"))
    P = lib.TreeParams.from_buffer_copy(bytes(r["P"]))
    states = torch.from_numpy(r["entry"].view(np.uint8).copy()).to(dev)
    jobs = np.zeros(1, np.dtype(lib.CTU_JOB_DTYPE))
    jobs["x"], jobs["y"] = r["x0"], r["y0"]
    jt = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    I, keep = None, []
    if r["slice_type"] != 2:
        dplanes = {}

        def addr_of(a, off):
            if id(a) not in dplanes:
                dplanes[id(a)] = torch.from_numpy(a.copy()).to(dev)
            return dplanes[id(a)].data_ptr() + 2 * off
        tab = refpic_table_of(r, addr_of)
        mv, mr = torch.from_numpy(m["mv"].copy()).to(dev), torch.from_numpy(m["refi"].copy()).to(dev)
        col = [torch.from_numpy(a.copy()).to(dev) for a in r["col"]]
        I = lib.TreeInter()
        I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, r["s_ref"][0], r["s_ref"][1], hip_params(r["ipar"])
        I.map_mv, I.map_refi, I.col_mv0, I.col_mv1, I.ecu_depth = mv.data_ptr(), mr.data_ptr(), col[0].data_ptr(), col[1].data_ptr(), r["ecu_depth"]
        keep = [tab, dplanes, col]
    out, nxt, _ = call([t.data_ptr() for t in org], org[0].shape[1], org[1].shape[1], [t.data_ptr() for t in mod], mod[0].shape[1], mod[1].shape[1], ms, mi, mt, mc, states, P, jt,
                       inter=I)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    after = dict(scu=ms.cpu().numpy().view(np.uint32), ipm=mi.cpu().numpy(), cu_mode=mc.cpu().numpy().view(np.uint32))
    if I is not None:
        after["mv"], after["refi"] = mv.cpu().numpy(), mr.cpu().numpy()
    del keep
    return (out.cpu().numpy().reshape(-1).view(CTU_DATA_DTYPE), nxt.cpu().numpy().reshape(-1).view(SBAC_DTYPE), after, [t.cpu().numpy() for t in mod])


def oracle_as_engine(r):
    """a stand-in for xeve_amd.device.mode_analyze_ctu_jobs that runs the oracle on the same operands (host memory): lets the CPU suite check run_walk's plumbing --
    pointers, strides, the reference-picture table, the records -- without a GPU"""
    import torch

    def call(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, ms, mi, mt, mc, states, P, jobs, inter=None):
        j = jobs.numpy().view(np.dtype([("x", "<i4"), ("y", "<i4"), ("sbac", "<i4"), ("pic", "<i4")]))[0]
        d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
        org, modp = (c_void_p * 3)(*org_ptrs), (c_void_p * 3)(*mod_ptrs)
        TP = TreeParams.from_buffer_copy(bytes(P))
        entry = states.numpy().reshape(-1).view(SBAC_DTYPE)
        if inter is None:
            oracle_tree().xo_mode_analyze_ctu_intra(org, s_org_l, s_org_c, modp, s_mod_l, s_mod_c, ms.data_ptr(), mi.data_ptr(), mt.data_ptr(), mc.data_ptr(), ptr(entry), C.byref(TP),
                                                    int(j["x"]), int(j["y"]), ptr(d), ptr(nb))
        else:
            I = TreeInter()
            I.refp, I.s_ref_l, I.s_ref_c, I.ipar = inter.refp, inter.s_ref_l, inter.s_ref_c, r["ipar"]  # (the oracle's own parameter layout; the library's is converted by hip_params)
            I.map_mv, I.map_refi, I.col0, I.col1, I.ecu_depth = inter.map_mv, inter.map_refi, inter.col_mv0, inter.col_mv1, inter.ecu_depth
            oracle_tree_any().xo_mode_analyze_ctu(org, s_org_l, s_org_c, modp, s_mod_l, s_mod_c, ms.data_ptr(), mi.data_ptr(), mt.data_ptr(), mc.data_ptr(), ptr(entry), C.byref(TP),
                                                  C.byref(I), int(j["x"]), int(j["y"]), ptr(d), ptr(nb))
        return torch.from_numpy(d.view(np.uint8).copy()), torch.from_numpy(nb.view(np.uint8).copy()), None
    return call


WRITER_FIELDS = ("range", "code", "code_bits", "stacked_ff", "stacked_zero", "pending_byte", "is_pending_byte", "ctx")


def writer_inputs(r):
    """what the reference's writer had in front of it for the record's CTU: the CTU's data as the reference decided it, the maps as the decision left them (coded flags of
    the CTU reset), the entry state = the state the decision entered with (it is the writer's: xeve_enc.c:139)"""
    maps = dict(scu=r["ref_maps"]["scu"].copy(), ipm=r["ref_maps"]["ipm"].copy(), tidx=r["maps"]["tidx"].copy(), cu_mode=r["ref_maps"]["cu_mode"].copy())
    state = r["entry"].copy()
    state["bitcounter"] = 0
    return r["ref_ctu"].copy(), state, maps


def writer_same_as_reference(r, state, out_bytes, maps):
    w, what = r["wr"], ("record", r["k"], r["clip"], "poc", r["poc"], "writer")
    for f in WRITER_FIELDS:
        assert np.array_equal(state[f], w["state"][f]), (what, f)
    assert np.array_equal(out_bytes, w["bytes"]), (what, "bytes", len(out_bytes), len(w["bytes"]))
    ctu, w_scu = 1 << r["P"].log2_ctu, r["w"] // 4
    glob = np.array([(r["y0"] // 4 + j) * w_scu + r["x0"] // 4 + i for j in range(min(ctu, r["h"] - r["y0"]) // 4) for i in range(min(ctu, r["w"] - r["x0"]) // 4)])
    assert np.array_equal(maps["scu"][glob], w["scu"][glob]) and np.array_equal(maps["cu_mode"][glob], w["cu_mode"][glob]), (what, "unit flags")
