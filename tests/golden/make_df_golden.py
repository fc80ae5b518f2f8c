#!/usr/bin/env python3
"""Writes tests/golden/df_v1.npz: inputs and outputs of the reference's in-loop deblocking (xeve_deblock & co., src_base/xeve_df.c,
both edge directions as xeve_loop_filter runs them) and of xeve_picbuf_expand, via oracle/ref_df_driver.c.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _df_cases import PAD, make_case, origin  # noqa: E402
from _libs import ptr, ref_df  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "df_v1.npz")
R = ref_df()
r = np.random.default_rng(90125)
d, k = {}, 0
for (w, h, bd, idc, min_cu) in [(128, 64, 10, 1, 4), (200, 136, 10, 1, 8), (64, 64, 8, 1, 4), (96, 72, 10, 0, 4), (72, 40, 12, 3, 4), (256, 128, 10, 1, 8)]:
    c = make_case(r, w, h, bd, idc, min_cu)
    out = [p.copy() for p in c["planes"]]
    ms, cm = c["map_scu"].copy(), c["map_cu_mode"].copy()
    R.refdrv_deblock_picture(ptr(out[0], origin(c, 0)), ptr(out[1], origin(c, 1)), ptr(out[2], origin(c, 2)), c["s_l"], c["s_c"], ptr(ms), ptr(cm),
                             ptr(c["refi"]), ptr(c["mv"]), c["p"])
    for i in range(3):
        d["in%d_%d" % (k, i)], d["out%d_%d" % (k, i)] = c["planes"][i], out[i]
    d["map_scu%d" % k], d["map_cu_mode%d" % k], d["refi%d" % k], d["mv%d" % k] = c["map_scu"], c["map_cu_mode"], c["refi"], c["mv"]
    d["p%d" % k] = np.frombuffer(bytes(c["p"]), dtype=np.int32).copy()
    k += 1
d["n"] = np.array(k)
# padding: one plane, three geometries
for j, (w, h, e) in enumerate([(64, 32, 16), (40, 24, 9), (8, 8, 16)]):
    s = w + 2 * PAD
    a = r.integers(0, 1024, size=(h + 2 * PAD, s)).astype(np.int16)
    b = a.copy()
    z = np.zeros(4, np.int16)
    R.refdrv_picbuf_expand(ptr(b, PAD * s + PAD), ptr(z), ptr(z), s, 0, w, h, 0, 0, e, 0, 0)
    d["pad_in%d" % j], d["pad_out%d" % j], d["pad_p%d" % j] = a, b, np.array([w, h, e, s])
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), k)
