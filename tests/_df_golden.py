"""Iterator over tests/golden/df_v1.npz (reference outputs of the in-loop deblocking and of xeve_picbuf_expand)."""
import os

import numpy as np

from _df_cases import PAD
from _libs import DeblockParams

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "df_v1.npz")


def golden():
    g = np.load(GOLD)
    for k in range(int(g["n"])):
        p = DeblockParams.from_buffer_copy(np.ascontiguousarray(g["p%d" % k]).tobytes())
        idc = p.chroma_format_idc
        ws = 1 if idc <= 2 else 0
        yield dict(planes=[np.ascontiguousarray(g["in%d_%d" % (k, i)]) for i in range(3)], out=[g["out%d_%d" % (k, i)] for i in range(3)],
                   s_l=p.w + 2 * PAD, s_c=(p.w >> ws) + 2 * PAD, map_scu=np.ascontiguousarray(g["map_scu%d" % k]),
                   map_cu_mode=np.ascontiguousarray(g["map_cu_mode%d" % k]), refi=np.ascontiguousarray(g["refi%d" % k]),
                   mv=np.ascontiguousarray(g["mv%d" % k]), p=p)


def golden_pad():
    g = np.load(GOLD)
    for j in range(3):
        w, h, e, s = (int(v) for v in g["pad_p%d" % j])
        yield np.ascontiguousarray(g["pad_in%d" % j]), g["pad_out%d" % j], w, h, e, s
