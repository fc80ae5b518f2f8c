"""Host-side mirror of the reference's dispatch interface for the hot path, backed by the HIP tables.

Names follow the reference (src_base/xeve_sad.h:52-64, xeve_mc.h:92-104, xeve_tq.h:63, xeve_type.h:984):
``func_sad[log2w][log2h](w, h, src1, src2, s_src1, s_src2, bit_depth)`` etc.  Arguments are numpy int16
arrays plus element offsets (a block inside a plane = ``(array, offset)``), strides in elements -- the same
meaning as the reference's raw pointers.  Used by the parity tests so that they read like calls on the
reference's tables; nothing here computes anything on the CPU.
"""
import ctypes as C

import numpy as np

from . import lib as _lib


def _p(a, off=0):
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


class HipTables:
    def __init__(self, device=0):
        _lib.init(device)
        L = _lib.load()
        t = L.tables
        self.func_sad = [[t["xeve_tbl_sad_16b_hip"][i * 8 + j] for j in range(8)] for i in range(8)]
        self.func_ssd = [[t["xeve_tbl_ssd_16b_hip"][i * 8 + j] for j in range(8)] for i in range(8)]
        self.func_diff = [[t["xeve_tbl_diff_16b_hip"][i * 8 + j] for j in range(8)] for i in range(8)]
        self.func_satd = [t["xeve_tbl_satd_16b_hip"][0]]
        self.func_mc_l = [[t["xeve_tbl_mc_l_hip"][i * 2 + j] for j in range(2)] for i in range(2)]
        self.func_mc_c = [[t["xeve_tbl_mc_c_hip"][i * 2 + j] for j in range(2)] for i in range(2)]
        self.func_txb = list(t["xeve_tbl_txb_hip"])
        self.fn_itxb = list(t["xeve_tbl_itxb_hip"])
        # Main profile, first slice (src_main/xevem_mc.h:52-54, xevem_tq.h:51, xevem_itdq.c:39); raw XEVEM_MC / XEVE_TX entries: call them with ctypes pointers
        self.func_dmvr_mc_l = [[t["xevem_tbl_dmvr_mc_l_hip"][i * 2 + j] for j in range(2)] for i in range(2)]
        self.func_dmvr_mc_c = [[t["xevem_tbl_dmvr_mc_c_hip"][i * 2 + j] for j in range(2)] for i in range(2)]
        self.func_bl_mc_l = [[t["xevem_tbl_bl_mc_l_hip"][i * 2 + j] for j in range(2)] for i in range(2)]
        self.func_tx = list(t["xeve_tbl_tx_hip"])
        self.func_itx = list(t["xeve_tbl_itx_hip"])
        self.func_itrans = [[t["xeve_itrans_map_tbl_hip"][i * 5 + j] for j in range(5)] for i in range(16)]  # xeve_func_itrans[type][log2 N - 1] (xevem_itdq.c:51)
        self.func_aff_h_sobel_flt, self.func_aff_v_sobel_flt = L.xevem_scaled_horizontal_sobel_filter_hip, L.xevem_scaled_vertical_sobel_filter_hip
        self.func_aff_eq_coef_comp = L.xevem_equal_coeff_computer_hip
        self.func_intra_pred_ang = [[t["xeve_tbl_intra_pred_ang_hip"][i * 2 + j] for j in range(2)] for i in range(3)]
        self.func_average_no_clip = L.xeve_average_16b_no_clip_hip
        self.fn_recon = L.xeve_recon_blk_hip

    # the reference's call macros (xeve_sad.h:57-64, xeve_mc.h:96-104), on (array, offset) operands ----
    @staticmethod
    def _l2(v):
        return int(v).bit_length() - 1

    def sad(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.func_sad[self._l2(w)][self._l2(h)](w, h, _p(a, oa), _p(b, ob), s1, s2, bd)

    def ssd(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.func_ssd[self._l2(w)][self._l2(h)](w, h, _p(a, oa), _p(b, ob), s1, s2, bd)

    def satd(self, w, h, a, oa, b, ob, s1, s2, bd):
        return self.func_satd[0](w, h, _p(a, oa), _p(b, ob), s1, s2, bd)

    def diff(self, w, h, a, oa, b, ob, s1, s2, s_diff, out, bd):
        self.func_diff[self._l2(w)][self._l2(h)](w, h, _p(a, oa), _p(b, ob), s1, s2, s_diff, _p(out), bd)

    def mc_l(self, frac_x, frac_y, ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bd, coef):
        self.func_mc_l[int(frac_x != 0)][int(frac_y != 0)](_p(ref), gmv_x, gmv_y, s_ref, s_pred, _p(pred), w, h, bd, _p(coef))

    def mc_c(self, frac_x, frac_y, ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bd, coef):
        self.func_mc_c[int(frac_x != 0)][int(frac_y != 0)](_p(ref), gmv_x, gmv_y, s_ref, s_pred, _p(pred), w, h, bd, _p(coef))

    def avg(self, a, b, d, sa, sb, sd, w, h):
        self.func_average_no_clip(_p(a), _p(b), _p(d), sa, sb, sd, w, h)

    def tx(self, log2n, src, dst, shift, line, step):
        self.func_txb[log2n - 1](_p(src), _p(dst), shift, line, step)

    def itx(self, log2n, src, dst, shift, line, step):
        self.fn_itxb[log2n - 1](_p(src), _p(dst), shift, line, step)

    def trans(self, coef, log2w, log2h, bd):
        """xeve_trans (xeve_tq.c:396-404) driven through the two table calls, as the reference does."""
        tb = np.zeros(1 << (log2w + log2h), np.int32)
        self.tx(log2w, coef, tb, 0, 1 << log2h, 0)
        self.tx(log2h, tb, coef, (log2w - 1 + bd - 8) + (log2h + 6), 1 << log2w, 1)

    def itrans(self, coef, log2w, log2h, bd):
        """xeve_itrans (xeve_itdq.c:435-440)."""
        tb = np.zeros(1 << (log2w + log2h), np.int32)
        self.itx(log2h, coef, tb, 0, 1 << log2w, 0)
        self.itx(log2w, tb, coef, 7 + 12 - (bd - 8), 1 << log2h, 1)

    def recon(self, coef, pred, is_coef, cuw, cuh, s_rec, rec, bd):
        self.fn_recon(_p(coef), _p(pred), is_coef, cuw, cuh, s_rec, _p(rec), bd)
