"""In-loop deblocking and picture padding on the GPU (xeve_hip_deblock, xeve_hip_picbuf_expand) against the reference goldens and
the pinned oracle, through the C-ABI."""
import numpy as np
import pytest

from _df_cases import PAD, make_case, origin, tile_map
from _df_golden import golden, golden_pad
from _libs import oracle_df, ptr

pytestmark = pytest.mark.gpu


def run_hip(c, tidx=None):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    planes = [torch.from_numpy(p.copy()).to(dev) for p in c["planes"]]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    D.deblock(planes, [origin(c, k) for k in range(3)], c["s_l"], c["s_c"], up(c["map_scu"]), up(c["map_cu_mode"]), up(c["refi"]), up(c["mv"]),
              lib.DeblockParams.from_buffer_copy(bytes(c["p"])), map_tidx=None if tidx is None else up(tidx))
    torch.cuda.synchronize()
    return [p.cpu().numpy() for p in planes]


def test_hip_deblock_matches_reference_goldens():
    n = 0
    for c in golden():
        got = run_hip(c)
        for k in range(3):
            assert np.array_equal(got[k], c["out"][k]), (n, k, np.argwhere(got[k] != c["out"][k])[:4])
        n += 1
    assert n == 6


@pytest.mark.parametrize("w,h,bd,idc,min_cu", [(256, 192, 10, 1, 4), (512, 320, 10, 1, 8), (136, 72, 8, 1, 4), (192, 64, 12, 3, 4), (320, 200, 10, 0, 4),
                                               (1920, 1080, 10, 1, 8)])
def test_hip_deblock_vs_oracle(w, h, bd, idc, min_cu):
    O = oracle_df()
    r = np.random.default_rng(w + 3 * h + bd + idc)
    for rep in range(2 if w < 1000 else 1):
        c = make_case(r, w, h, bd, idc, min_cu)
        e = [p.copy() for p in c["planes"]]
        ms = c["map_scu"].copy()
        O.xo_deblock_picture(ptr(e[0], origin(c, 0)), ptr(e[1], origin(c, 1)), ptr(e[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms),
                             ptr(c["map_cu_mode"]), ptr(c["refi"]), ptr(c["mv"]), c["p"])
        got = run_hip(c)
        for k in range(3):
            assert np.array_equal(got[k], e[k]), (rep, k, np.argwhere(got[k] != e[k])[:4])


@pytest.mark.parametrize("w,h,min_cu,sx,sy", [(256, 128, 8, 2, 0), (320, 200, 4, 3, 2), (1920, 1080, 8, 15, 8)])
def test_hip_deblock_with_tiles_vs_oracle(w, h, min_cu, sx, sy):
    """edges between units of different tiles stay unfiltered (ctx->map_tidx; the oracle's tile path is pinned to the reference's per-tile loop)"""
    O = oracle_df()
    r = np.random.default_rng(w + h + sx + sy)
    c = make_case(r, w, h, 10, 1, min_cu)
    tid = tile_map(c, sx, sy)
    e = [p.copy() for p in c["planes"]]
    ms = c["map_scu"].copy()
    O.xo_deblock_picture_tiles(ptr(e[0], origin(c, 0)), ptr(e[1], origin(c, 1)), ptr(e[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms), ptr(c["map_cu_mode"]), ptr(tid),
                               ptr(c["refi"]), ptr(c["mv"]), c["p"])
    got = run_hip(c, tid)
    for k in range(3):
        assert np.array_equal(got[k], e[k]), (k, np.argwhere(got[k] != e[k])[:4])
    one = run_hip(c)
    assert any(not np.array_equal(one[k], got[k]) for k in range(3))


def test_hip_deblock_runs_of_4x4_cus():
    """every CU 4x4: each chroma edge reads what its neighbour wrote, along whole rows and columns (the serial chains)"""
    O = oracle_df()
    r = np.random.default_rng(99)
    c = make_case(r, 128, 128, 10, 1, 4)
    lg = 2
    c["map_cu_mode"][:] = (lg << 24) | (lg << 28)
    e = [p.copy() for p in c["planes"]]
    ms = c["map_scu"].copy()
    O.xo_deblock_picture(ptr(e[0], origin(c, 0)), ptr(e[1], origin(c, 1)), ptr(e[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms), ptr(c["map_cu_mode"]),
                         ptr(c["refi"]), ptr(c["mv"]), c["p"])
    got = run_hip(c)
    for k in range(3):
        assert np.array_equal(got[k], e[k]), k


def test_hip_picbuf_expand():
    import torch

    import xeve_amd
    from xeve_amd import device as D

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    for a, out, w, h, e, s in golden_pad():
        planes = [torch.from_numpy(a.copy()).to(dev) for _ in range(3)]
        o = PAD * s + PAD
        D.picbuf_expand(planes, [o, o, o], s, s, w, h, w, h, e, e, 1)
        for p in planes:
            assert np.array_equal(p.cpu().numpy(), out)
    # a 4:2:0 picture with the reference's padding depths (144 / 72), against the oracle
    O = oracle_df()
    r = np.random.default_rng(7)
    w, h, el, ec = 320, 192, 144, 72
    sl, sc = w + 2 * el, w // 2 + 2 * ec
    pl = [r.integers(0, 1024, size=(h + 2 * el, sl)).astype(np.int16), r.integers(0, 1024, size=(h // 2 + 2 * ec, sc)).astype(np.int16),
          r.integers(0, 1024, size=(h // 2 + 2 * ec, sc)).astype(np.int16)]
    org = [el * sl + el, ec * sc + ec, ec * sc + ec]
    dv = [torch.from_numpy(p.copy()).to(dev) for p in pl]
    D.picbuf_expand(dv, org, sl, sc, w, h, w // 2, h // 2, el, ec, 1)
    O.xo_picbuf_expand(ptr(pl[0], org[0]), sl, w, h, el)
    O.xo_picbuf_expand(ptr(pl[1], org[1]), sc, w // 2, h // 2, ec)
    O.xo_picbuf_expand(ptr(pl[2], org[2]), sc, w // 2, h // 2, ec)
    for k in range(3):
        assert np.array_equal(dv[k].cpu().numpy(), pl[k]), k
