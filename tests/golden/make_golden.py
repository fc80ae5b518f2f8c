#!/usr/bin/env python3
"""Generates tests/golden/hotpath_v1.npz from the UNMODIFIED reference compiled in place (oracle/_ref).

Run in the build container only (needs oracle/_ref/libxeveb_ref.so, i.e. /root/reference):
    python tests/golden/make_golden.py
Each case stores the inputs and the outputs of the reference's plain-C dispatch tables
(xeve_tbl_sad_16b, xeve_tbl_ssd_16b, xeve_tbl_diff_16b, xeve_tbl_satd_16b, xeve_tbl_mc_l/_c,
xeve_average_16b_no_clip, xeve_tbl_txb, xeve_tbl_itxb, xeve_recon_blk).  The fixture is data only.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _libs import ilog2, ptr, ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_v1.npz")


def main():
    R = ref()
    assert R is not None, "oracle/_ref missing: run `make -C oracle ref` first"
    V = R.variants["c"]
    r = np.random.default_rng(20250905)
    d = {}
    meta = []  # rows: kind, index, then parameters

    def pels(shape, bd):
        return r.integers(0, 1 << bd, size=shape, dtype=np.int16)

    # --- sad / ssd / satd / diff: baseline shapes + a few rectangles + tiny ------------------------------
    shapes = [(8, 8), (16, 16), (32, 32), (64, 64), (4, 4), (2, 2), (16, 8), (8, 16), (8, 4), (4, 8), (32, 8), (64, 16),
              (128, 128), (1, 1), (4, 64)]
    k = 0
    for bd in (8, 10):
        for (w, h) in shapes:
            s1, s2 = w + int(r.integers(0, 7)), w + int(r.integers(0, 7))
            a, b = pels((h, s1), bd), pels((h, s2), bd)
            if k % 5 == 4:
                a = (2 * a.astype(np.int32) - pels((h, s1), bd)).astype(np.int16)  # org_bi range
            lw, lh = ilog2(w), ilog2(h)
            sad = V.sad[lw * 8 + lh](w, h, ptr(a), ptr(b), s1, s2, bd)
            ssd = V.ssd[lw * 8 + lh](w, h, ptr(a), ptr(b), s1, s2, bd)
            satd = V.satd[0](w, h, ptr(a), ptr(b), s1, s2, bd) if (w % 2 == 0 and h % 2 == 0) else -1
            df = np.zeros((h, w), np.int16)
            V.diff[lw * 8 + lh](w, h, ptr(a), ptr(b), s1, s2, w, ptr(df), bd)
            d["dist%d_a" % k], d["dist%d_b" % k], d["dist%d_diff" % k] = a, b, df
            d["dist%d_p" % k] = np.array([w, h, s1, s2, bd, sad, ssd, satd], np.int64)
            k += 1
    d["dist_n"] = np.array(k)

    # --- mc luma / chroma: every phase, two sizes each --------------------------------------------------
    k = 0
    for luma in (1, 0):
        taps, unit = (8, 16) if luma else (4, 32)
        step, nph = 4, (4 if luma else 8)
        tbl = V.mc_l if luma else V.mc_c
        coef = R.mc_l_coeff if luma else R.mc_c_coeff
        for (w, h) in ([(8, 8), (32, 16), (64, 64)] if luma else [(4, 4), (16, 8), (32, 32)]):
            pad = 8
            s_ref = w + 2 * pad + 3
            plane = pels((h + 2 * pad, s_ref), 10)
            for px in range(nph):
                for py in range(nph):
                    dx, dy = px * step, py * step
                    gx, gy = pad * unit + dx, pad * unit + dy
                    out = np.zeros((h, w), np.int16)
                    tbl[(dx != 0) * 2 + (dy != 0)](ptr(plane), gx, gy, s_ref, w, ptr(out), w, h, 10, coef)
                    d["mc%d_ref" % k], d["mc%d_out" % k] = plane, out
                    d["mc%d_p" % k] = np.array([luma, w, h, s_ref, gx, gy, dx, dy, 10], np.int64)
                    k += 1
    d["mc_n"] = np.array(k)
    d["mc_l_coeff"] = np.frombuffer(R.mc_l_coeff, dtype=np.int16).reshape(16, 8).copy()
    d["mc_c_coeff"] = np.frombuffer(R.mc_c_coeff, dtype=np.int16).reshape(32, 4).copy()

    # --- average ------------------------------------------------------------------------------------------
    a, b = pels((16, 16), 10), pels((16, 16), 10)
    o = np.zeros((16, 16), np.int16)
    V.avg(ptr(a), ptr(b), ptr(o), 16, 16, 16, 16, 16)
    d["avg_a"], d["avg_b"], d["avg_out"] = a, b, o

    # --- 2-D transforms through the two table calls (xeve_tq.c:396-404, xeve_itdq.c:435-440) -------------
    k = 0
    for bd in (8, 10):
        for (lw, lh) in [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 5), (6, 4), (2, 6)]:
            n = 1 << (lw + lh)
            resid = r.integers(-(1 << bd) + 1, 1 << bd, size=n, dtype=np.int16)
            c = resid.copy()
            tb = np.zeros(n, np.int32)
            V.txb[lw - 1](ptr(c), ptr(tb), 0, 1 << lh, 0)
            V.txb[lh - 1](ptr(tb), ptr(c), (lw - 1 + bd - 8) + (lh + 6), 1 << lw, 1)
            fwd = c.copy()
            V.itxb[lh - 1](ptr(c), ptr(tb), 0, 1 << lw, 0)
            V.itxb[lw - 1](ptr(tb), ptr(c), 7 + 12 - (bd - 8), 1 << lh, 1)
            d["tq%d_in" % k], d["tq%d_fwd" % k], d["tq%d_inv" % k] = resid, fwd, c.copy()
            d["tq%d_p" % k] = np.array([lw, lh, bd], np.int64)
            k += 1
    d["tq_n"] = np.array(k)
    for n in (2, 4, 8, 16, 32, 64):
        d["tm%d" % n] = R.tm(n)

    # --- recon ---------------------------------------------------------------------------------------------
    coef = r.integers(-2048, 2048, size=32 * 32, dtype=np.int16)
    pred = pels(32 * 32, 10)
    for is_coef in (0, 1):
        rec = np.zeros((32, 40), np.int16)
        R.recon(ptr(coef), ptr(pred), is_coef, 32, 32, 40, ptr(rec), 10)
        d["recon%d" % is_coef] = rec
    d["recon_coef"], d["recon_pred"] = coef, pred

    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
