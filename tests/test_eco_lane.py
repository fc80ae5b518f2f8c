"""The lane-serial bitstream writer of a decided CTU (xeve_amd/csrc/eco_lane.h: xeve_eco_tree as one GPU lane runs it for one chain) against the oracle's xo_eco_ctu, which
is pinned beside the reference's writer in the live encoder (test_integration_ref.py, shadow mode).  On the CPU: the header's functions are __host__ __device__.
Whole small pictures, I and P / B: the oracle decides every CTU from the WRITER's state (the true chain: xeve_enc.c:139), both writers write it -- coder state after
each CTU field for field, the bytes emitted, the unit flags xeve_eco_unit stores."""
import ctypes as C

import numpy as np
import pytest

import _lane
from _libs import SBAC_DTYPE, c_int, c_void_p, oracle, ptr
from _tree_cases import CASES, CTU_DATA_DTYPE, INTER_CASES, TreeInter, TreeParams, make_case, make_inter_case, oracle_tree, oracle_tree_any
from _mc_cases import refpic_table

pytestmark = pytest.mark.skipif(not _lane.available(), reason="hipcc not found")


def oracle_eco():
    O = oracle()
    O.xo_eco_ctu.restype = c_int
    O.xo_eco_ctu.argtypes = [c_void_p, c_void_p, C.POINTER(TreeParams), c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_void_p, c_int]
    return O


def write_both(P, nref, d, state, m_scu, m_ipm, m_tidx, m_cum, x, y, what):
    """m_scu: as the decision left it (coded flags set).  Returns the writer's state after the CTU; the maps are updated in place"""
    OE, L = oracle_eco(), _lane.lane()
    sc = _lane.scans()
    # oracle: coded flags of the CTU reset first (mode_analyze_lcu's tail, xeve_mode.c:2591-2607)
    o_scu, o_cum, o_state, o_bytes = m_scu.copy(), m_cum.copy(), state.copy(), np.zeros(1 << 16, np.uint8)
    ctu, w_scu = 1 << P.log2_ctu, P.ip.w_scu
    for j in range(min(ctu, P.pic_h - y) // 4):
        g = (y // 4 + j) * w_scu + x // 4
        o_scu[g:g + min(ctu, P.pic_w - x) // 4] &= np.uint32(0x7FFFFFFF)
    nr = (c_int * 2)(*nref)
    n_o = OE.xo_eco_ctu(ptr(o_state), ptr(d), C.byref(P), nr, ptr(o_scu), ptr(m_ipm), ptr(m_tidx), ptr(o_cum), x, y, ptr(o_bytes), o_bytes.size)
    l_scu, l_cum, l_state, l_bytes = m_scu.copy(), m_cum.copy(), state.copy(), np.zeros(1 << 16, np.uint8)
    n_l = L.xl_host_eco_ctu(P.ip.chroma_format_idc, P.ip.slice_type, P.log2_ctu, P.pic_w, P.pic_h, w_scu, nref[0], nref[1], ptr(sc[0]), ptr(sc[1]), ptr(sc[2]), ptr(l_state), ptr(d),
                            ptr(l_scu), ptr(m_ipm), ptr(m_tidx), ptr(l_cum), x, y, ptr(l_bytes), l_bytes.size)
    for f in ("range", "code", "code_bits", "stacked_ff", "stacked_zero", "pending_byte", "is_pending_byte", "bin_counter", "ctx"):
        assert np.array_equal(l_state[f], o_state[f]), (what, f, l_state[f], o_state[f])
    assert n_l == n_o and np.array_equal(l_bytes[:n_l], o_bytes[:n_o]), (what, "bytes", n_l, n_o)
    assert np.array_equal(l_scu, o_scu) and np.array_equal(l_cum, o_cum), (what, "unit flags")
    m_scu[:], m_cum[:] = o_scu, o_cum
    return o_state, n_o


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_lane_writer_on_i_pictures(case):
    c = make_case(*case)
    O = oracle_tree()
    total = 0
    for p in range(c["npic"]):
        org = (c_void_p * 3)(*[a[p].ctypes.data for a in c["org"]])
        mod = (c_void_p * 3)(*[a[p].ctypes.data for a in c["mod"]])
        m = c["maps"]
        state = c["entry"][p:p + 1].copy()
        state["bitcounter"] = 0  # (the writer's coder does not count)
        for (x, y) in c["order"]:
            d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
            O.xo_mode_analyze_ctu_intra(org, c["org"][0].shape[2], c["org"][1].shape[2], mod, c["mod"][0].shape[2], c["mod"][1].shape[2], ptr(m["scu"][p]), ptr(m["ipm"][p]),
                                        ptr(m["tidx"][p]), ptr(m["cu_mode"][p]), ptr(state), C.byref(c["P"]), x, y, ptr(d), ptr(nb))
            state, n = write_both(c["P"], (0, 0), d, state, m["scu"][p], m["ipm"][p], m["tidx"][p], m["cu_mode"][p], x, y, (case, p, x, y))
            total += n
    assert total > 100  # bytes did come out


@pytest.mark.parametrize("case", INTER_CASES, ids=[str(c[0]) for c in INTER_CASES])
def test_lane_writer_on_p_and_b_pictures(case):
    c = make_inter_case(*case)
    O = oracle_tree_any()
    refs, org = c["refs"], c["org"]
    tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    I = TreeInter()
    m = c["maps"]
    I.refp, I.s_ref_l, I.s_ref_c, I.ipar = tab.ctypes.data, refs["s_l"], refs["s_c"], c["ipar"]
    I.map_mv, I.map_refi, I.col0, I.col1, I.ecu_depth = m["mv"].ctypes.data, m["refi"].ctypes.data, c["col"][0].ctypes.data, c["col"][1].ctypes.data, c["ecu_depth"]
    orgp = (c_void_p * 3)(int(org[0].ctypes.data) + 2 * refs["org_l"], int(org[1].ctypes.data) + 2 * refs["org_c"], int(org[2].ctypes.data) + 2 * refs["org_c"])
    modp = (c_void_p * 3)(*[a.ctypes.data for a in c["mod"]])
    state = c["entry"][0:1].copy()
    state["bitcounter"] = 0
    nref = (c["ipar"].rdo.num_refp[0], c["ipar"].rdo.num_refp[1])
    total, modes = 0, set()
    for (x, y) in c["order"]:
        d, nb = np.zeros(1, CTU_DATA_DTYPE), np.zeros(1, SBAC_DTYPE)
        O.xo_mode_analyze_ctu(orgp, refs["s_l"], refs["s_c"], modp, c["mod"][0].shape[1], c["mod"][1].shape[1], ptr(m["scu"]), ptr(m["ipm"]), ptr(m["tidx"]), ptr(m["cu_mode"]),
                              ptr(state), C.byref(c["P"]), C.byref(I), x, y, ptr(d), ptr(nb))
        state, n = write_both(c["P"], nref, d, state, m["scu"], m["ipm"], m["tidx"], m["cu_mode"], x, y, (case, x, y))
        total += n
        modes |= set(np.unique(d["pred_mode"][0]).tolist())
    assert total > 10 and len(modes) >= 2  # bytes did come out; more than one kind of CU was written


def test_lane_tile_end_matches_oracle():
    """xeve_eco_tile_end_flag(1) + xeve_sbac_finish from many coder states (the states the writer tests above end in, and random ones): bytes and the state left behind"""
    from _sbac_cases import make_states

    O, L = oracle(), _lane.lane()
    O.xo_eco_tile_end.restype = c_int
    O.xo_eco_tile_end.argtypes = [c_void_p, c_void_p, c_int]
    r = np.random.default_rng(12)
    st = make_states(r, 400)
    st["bitcounter"] = 0
    st["code"] &= (1 << 19) - 1
    st["pending_byte"] = r.integers(0, 256, size=len(st))
    st["is_pending_byte"] = r.integers(0, 2, size=len(st))
    st["stacked_zero"] = r.integers(0, 3, size=len(st)) * st["is_pending_byte"]
    st["pending_byte"][::7] = 0
    st["code_bits"] = r.integers(1, 12, size=len(st))
    kinds = set()
    for i in range(len(st)):
        a, b = st[i:i + 1].copy(), st[i:i + 1].copy()
        ba, bb = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        na, nb = O.xo_eco_tile_end(ptr(a), ptr(ba), 64), L.xl_host_eco_tile_end(ptr(b), ptr(bb), 64)
        assert na == nb and np.array_equal(ba[:na], bb[:nb]), (i, na, nb, ba[:8], bb[:8])
        for f in ("range", "code", "code_bits", "stacked_ff", "stacked_zero", "pending_byte", "is_pending_byte"):
            assert a[f][0] == b[f][0], (i, f)
        kinds.add(na)
    assert len(kinds) >= 3


def test_lane_writer_on_more_pictures():
    """more seeds: 4:4:4 and 4:0:0 I pictures with partial CTUs, B pictures with two reference pictures per list and both early-termination depths"""
    for case in [(3201, 2, 104, 72, 10, 3, 6, 64, 4, 28), (3202, 2, 72, 104, 8, 0, 6, 32, 8, 36), (3203, 1, 96, 96, 10, 1, 5, 32, 4, 24)]:
        test_lane_writer_on_i_pictures(case)
    for case in [(4401, 136, 72, 10, 1, 0, 2, 1, 0.0), (4402, 128, 128, 10, 1, 1, 2, 0, 0.0), (4403, 72, 136, 10, 1, 0, 1, 0, 4.0)]:
        test_lane_writer_on_p_and_b_pictures(case)
