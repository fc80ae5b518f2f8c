"""Seeded deblocking cases: a picture, a random quad-tree partition per 64x64 CTU and the per-4x4 maps the reference keeps
(map_scu bit fields, map_cu_mode CU sizes, map_refi, map_mv)."""
import numpy as np

from _libs import DeblockParams

PAD = 16


def make_case(r, w, h, bd=10, idc=1, min_cu=4, log2_ctu=6):
    ws, hs = (1 if idc <= 2 else 0), (1 if idc <= 1 else 0)
    w_scu, h_scu = (w + 3) // 4, (h + 3) // 4
    s_l, s_c = w + 2 * PAD, (w >> ws) + 2 * PAD
    hl, hc = h + 2 * PAD, (h >> hs) + 2 * PAD
    maxv = (1 << bd) - 1

    def plane(hh, ss):  # blocky content + noise: plenty of edges the filter acts on, plenty it leaves alone
        base = np.repeat(np.repeat(r.integers(0, maxv + 1, size=((hh + 7) // 8, (ss + 7) // 8)), 8, axis=0), 8, axis=1)[:hh, :ss]
        return np.clip(base + r.integers(-12, 13, size=(hh, ss)), 0, maxv).astype(np.int16)

    planes = [plane(hl, s_l), plane(hc, s_c), plane(hc, s_c)]
    map_scu = np.zeros(w_scu * h_scu, np.uint32)
    map_cu_mode = np.zeros(w_scu * h_scu, np.uint32)
    refi = np.zeros((w_scu * h_scu, 2), np.int8)
    mv = np.zeros((w_scu * h_scu, 2, 2), np.int16)
    ctu = 1 << log2_ctu

    def leaf(x, y, size):
        sx, sy, n = x // 4, y // 4, size // 4
        lg = size.bit_length() - 1
        intra, cbfl, qp = int(r.random() < 0.15), int(r.random() < 0.5), int(r.integers(18, 52))
        m = (int(r.integers(0, 1 << 15))) | (intra << 15) | (qp << 16) | (cbfl << 24)  # low bits: fields the filter must ignore
        rf = (-1, -1) if intra else [(0, -1), (-1, 0), (0, 0), (1, 0), (0, 1), (1, -1)][int(r.integers(0, 6))]
        v = r.integers(-9, 10, size=(2, 2)) if r.random() < 0.7 else r.integers(-2, 3, size=(2, 2))
        for j in range(sy, min(sy + n, h_scu)):
            for i in range(sx, min(sx + n, w_scu)):
                t = j * w_scu + i
                map_scu[t], map_cu_mode[t] = m, (lg << 24) | (lg << 28)
                refi[t], mv[t] = rf, v

    def node(x, y, size):
        must = x + size > w or y + size > h
        if size > min_cu and (must or r.random() < (0.85 if size > 16 else 0.45)):
            hs_ = size // 2
            for k in range(4):
                xs, ys = x + (k & 1) * hs_, y + (k >> 1) * hs_
                if xs < w and ys < h:
                    node(xs, ys, hs_)
        else:
            leaf(x, y, size)

    for cy in range(0, h, ctu):
        for cx in range(0, w, ctu):
            node(cx, cy, ctu)
    p = DeblockParams()
    p.w, p.h, p.w_scu, p.h_scu, p.log2_max_cuwh = w, h, w_scu, h_scu, log2_ctu
    p.bit_depth_luma = p.bit_depth_chroma = bd
    p.chroma_format_idc = idc
    p.qp_u_offset, p.qp_v_offset = int(r.integers(-4, 5)), int(r.integers(-4, 5))
    bc = bd - 8
    for c in range(2):  # ctx->qp_chroma_dynamic: identity below zero, a monotone mapping into 0..51 above (xeve_util.c:1838-1846)
        for j in range(100):
            q = j - 6 * bc
            p.qp_chroma[c][j] = q if q < 30 else min(51, 29 + ((q - 29) * (3 + c)) // 4)
    return dict(planes=planes, s_l=s_l, s_c=s_c, map_scu=map_scu, map_cu_mode=map_cu_mode, refi=refi, mv=mv, p=p, ws=ws, hs=hs)


def tile_map(case, split_x_lcu, split_y_lcu):
    """ctx->map_tidx of a picture cut into up to 2 x 2 tiles after split_x_lcu CTU columns / split_y_lcu CTU rows (0 = no cut), tiles numbered in raster
    order (xeve_set_tile_info)"""
    p = case["p"]
    per = (1 << p.log2_max_cuwh) // 4
    w_lcu, h_lcu = (p.w + (1 << p.log2_max_cuwh) - 1) >> p.log2_max_cuwh, (p.h + (1 << p.log2_max_cuwh) - 1) >> p.log2_max_cuwh
    sx = split_x_lcu if 0 < split_x_lcu < w_lcu else w_lcu
    sy = split_y_lcu if 0 < split_y_lcu < h_lcu else h_lcu
    t = np.zeros((p.h_scu, p.w_scu), np.uint8)
    n = 0
    for (y0, y1) in ((0, sy), (sy, h_lcu)):
        for (x0, x1) in ((0, sx), (sx, w_lcu)):
            if x0 >= x1 or y0 >= y1:
                continue
            t[y0 * per:y1 * per, x0 * per:x1 * per] = n
            n += 1
    return t.reshape(-1).copy()


def origin(case, c):
    """element offset of sample (0, 0) in plane c"""
    return PAD * (case["s_l"] if c == 0 else case["s_c"]) + PAD
