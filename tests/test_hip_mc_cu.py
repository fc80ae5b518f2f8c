"""a8 on the GPU: xeve_hip_mc_cu_jobs (clip + per-list interpolation of Y/U/V + identical-motion shortcut + bi average) against
the pinned oracle's xo_mc_cu, through the C-ABI."""
import numpy as np
import pytest

from _libs import oracle_mc_cu, ptr
from _mc_cases import make_jobs, make_refs, refpic_table

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,bd,idc,nref", [(128, 96, 10, 1, 2), (64, 64, 8, 1, 1), (96, 64, 10, 3, 2), (72, 48, 10, 0, 3), (256, 128, 12, 1, 2)])
def test_hip_mc_cu_vs_oracle(w, h, bd, idc, nref):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O = oracle_mc_cu()
    r = np.random.default_rng(2 * w + h + bd + idc)
    refs = make_refs(r, w, h, bd, nref, idc)
    host_tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    dplanes = [[torch.from_numpy(p).to(dev) for p in pic] for pic in refs["pics"]]
    lut = {id(p): t for pic, dp in zip(refs["pics"], dplanes) for p, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
    for (cuw, cuh) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (4, 4), (4, 8)]:
        if cuw > w or cuh > h:
            continue
        jobs = make_jobs(r, 150, w, h, cuw, cuh, nref)
        cw, ch = cuw >> refs["ws"], cuh >> refs["hs"]
        got = D.mc_cu_jobs(dev_tab, (nref, nref), refs["s_l"], refs["s_c"], w, h, torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), cuw, cuh, bd, idc)
        got = [g.cpu().numpy() for g in got]
        for i in range(len(jobs)):
            e = [np.zeros(cuw * cuh, np.int16), np.zeros(cw * ch, np.int16), np.zeros(cw * ch, np.int16)]
            O.xo_mc_cu(ptr(host_tab), refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), cuw, cuh, bd, bd, idc, ptr(e[0]), ptr(e[1]), ptr(e[2]))
            for k in range(3 if idc else 1):
                assert np.array_equal(got[k][i], e[k]), (cuw, cuh, i, k, jobs[i])


def test_hip_mc_cu_one_list_only():
    """P-slice shape: no reference pictures in list 1"""
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    O = oracle_mc_cu()
    r = np.random.default_rng(8)
    w, h, bd = 128, 64, 10
    refs = make_refs(r, w, h, bd, 2)
    host_tab = refpic_table(refs, lambda a, off: int(a.ctypes.data) + 2 * off)
    dplanes = [[torch.from_numpy(p).to(dev) for p in pic] for pic in refs["pics"]]
    lut = {id(p): t for pic, dp in zip(refs["pics"], dplanes) for p, t in zip(pic, dp)}
    dev_tab = refpic_table(refs, lambda a, off: lut[id(a)].data_ptr() + 2 * off).view(lib.REFPIC_DTYPE)
    jobs = make_jobs(r, 80, w, h, 16, 16, 2)
    jobs["refi"][:, 1] = -1
    jobs["refi"][:, 0] = np.maximum(jobs["refi"][:, 0], 0)
    got = [g.cpu().numpy() for g in D.mc_cu_jobs(dev_tab, (2, 0), refs["s_l"], refs["s_c"], w, h, torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), 16, 16, bd)]
    for i in range(len(jobs)):
        e = [np.zeros(256, np.int16), np.zeros(64, np.int16), np.zeros(64, np.int16)]
        O.xo_mc_cu(ptr(host_tab), refs["s_l"], refs["s_c"], w, h, ptr(jobs[i:i + 1]), 16, 16, bd, bd, 1, ptr(e[0]), ptr(e[1]), ptr(e[2]))
        for k in range(3):
            assert np.array_equal(got[k][i], e[k]), (i, k)
