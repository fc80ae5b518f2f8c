"""Runs any table backend (OracleTables / HipTables) over the committed golden vectors
(tests/golden/hotpath_v1.npz: inputs + outputs of the reference's own plain-C tables)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz")


def run_golden(T):
    g = np.load(GOLDEN)
    n_checked = 0
    for k in range(int(g["dist_n"])):
        w, h, s1, s2, bd, sad, ssd, satd = (int(v) for v in g["dist%d_p" % k])
        a, b = np.ascontiguousarray(g["dist%d_a" % k]), np.ascontiguousarray(g["dist%d_b" % k])
        assert T.sad(w, h, a, 0, b, 0, s1, s2, bd) == sad, ("sad", k, w, h)
        assert T.ssd(w, h, a, 0, b, 0, s1, s2, bd) == ssd, ("ssd", k, w, h)
        if satd >= 0:
            assert T.satd(w, h, a, 0, b, 0, s1, s2, bd) == satd, ("satd", k, w, h)
        out = np.zeros((h, w), np.int16)
        T.diff(w, h, a, 0, b, 0, s1, s2, w, out, bd)
        assert np.array_equal(out, g["dist%d_diff" % k]), ("diff", k)
        n_checked += 4
    cl, cc = np.ascontiguousarray(g["mc_l_coeff"]), np.ascontiguousarray(g["mc_c_coeff"])
    for k in range(int(g["mc_n"])):
        luma, w, h, s_ref, gx, gy, dx, dy, bd = (int(v) for v in g["mc%d_p" % k])
        plane = np.ascontiguousarray(g["mc%d_ref" % k])
        out = np.zeros((h, w), np.int16)
        (T.mc_l if luma else T.mc_c)(dx, dy, plane, gx, gy, s_ref, w, out, w, h, bd, cl if luma else cc)
        assert np.array_equal(out, g["mc%d_out" % k]), ("mc", k, luma, w, h, dx, dy)
        n_checked += 1
    a, b = np.ascontiguousarray(g["avg_a"]), np.ascontiguousarray(g["avg_b"])
    o = np.zeros((16, 16), np.int16)
    T.avg(a, b, o, 16, 16, 16, 16, 16)
    assert np.array_equal(o, g["avg_out"])
    for k in range(int(g["tq_n"])):
        lw, lh, bd = (int(v) for v in g["tq%d_p" % k])
        c = np.ascontiguousarray(g["tq%d_in" % k]).copy()
        T.trans(c, lw, lh, bd)
        assert np.array_equal(c, g["tq%d_fwd" % k]), ("trans", k, lw, lh)
        T.itrans(c, lw, lh, bd)
        assert np.array_equal(c, g["tq%d_inv" % k]), ("itrans", k, lw, lh)
        n_checked += 2
    coef, pred = np.ascontiguousarray(g["recon_coef"]), np.ascontiguousarray(g["recon_pred"])
    for is_coef in (0, 1):
        rec = np.zeros((32, 40), np.int16)
        T.recon(coef, pred, is_coef, 32, 32, 40, rec, 10)
        assert np.array_equal(rec, g["recon%d" % is_coef]), ("recon", is_coef)
    return n_checked
