"""Seeded cases for the CU motion-compensation driver (xeve_mc, xeve_mc.c:465-610): reference pictures with the reference's
padding, CU jobs whose vectors reach far outside the picture (clipping), all prediction directions, identical-motion pairs."""
import numpy as np

from _libs import CU_MC_JOB_DTYPE, REFPIC_DTYPE

PAD_L, PAD_C = 144, 72


def make_refs(r, w, h, bd, nref, idc=1):
    ws, hs = (1 if idc <= 2 else 0), (1 if idc <= 1 else 0)
    s_l, s_c = w + 2 * PAD_L, (w >> ws) + 2 * PAD_L  # (chroma padded as deep as luma: 4:4:4 vectors reach as far)
    maxv = (1 << bd) - 1
    pics = []
    for i in range(nref * 2):
        pics.append([r.integers(0, maxv + 1, size=(h + 2 * PAD_L, s_l)).astype(np.int16),
                     r.integers(0, maxv + 1, size=((h >> hs) + 2 * PAD_L, s_c)).astype(np.int16),
                     r.integers(0, maxv + 1, size=((h >> hs) + 2 * PAD_L, s_c)).astype(np.int16)])
    pocs = r.integers(0, 3, size=nref * 2) * 2  # few distinct values: equal POCs across the lists do occur
    return dict(pics=pics, pocs=pocs, s_l=s_l, s_c=s_c, org_l=PAD_L * s_l + PAD_L, org_c=PAD_L * s_c + PAD_L, ws=ws, hs=hs)


def refpic_table(refs, addr_of):
    """REFPIC_DTYPE array [refi * 2 + list]; addr_of(plane_array, element_offset) -> address (host or device)"""
    t = np.zeros(len(refs["pics"]), REFPIC_DTYPE)
    for i, p in enumerate(refs["pics"]):
        t["y"][i], t["u"][i], t["v"][i] = addr_of(p[0], refs["org_l"]), addr_of(p[1], refs["org_c"]), addr_of(p[2], refs["org_c"])
        t["poc"][i] = refs["pocs"][i]
    return t


def make_jobs(r, n, w, h, cuw, cuh, nref):
    j = np.zeros(n, CU_MC_JOB_DTYPE)
    j["x"] = r.integers(0, max(1, (w - cuw) // 4 + 1), size=n) * 4
    j["y"] = r.integers(0, max(1, (h - cuh) // 4 + 1), size=n) * 4
    kind = r.integers(0, 4, size=n)  # 0 L0, 1 L1, 2 BI, 3 BI with the same vector (identical motion when the POCs agree)
    far = r.random(n) < 0.35
    mv = np.where(far[:, None, None], r.integers(-900, 901, size=(n, 2, 2)), r.integers(-40, 41, size=(n, 2, 2)))
    mv[kind == 3, 1] = mv[kind == 3, 0]
    j["mv"] = mv
    j["refi"][:, 0] = np.where(kind == 1, -1, r.integers(0, nref, size=n))
    j["refi"][:, 1] = np.where(kind == 0, -1, r.integers(0, nref, size=n))
    return j
