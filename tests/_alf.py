"""Helpers of the adaptive-loop-filter tests (Main profile): seeded cases, the oracle's restatement (oracle/xeve_oracle.c xo_alf_*), the reference's own
functions called in place (oracle/_ref/libxevem_ref.so: alf_derive_classification_blk, alf_filter_blk_7 / _5, xeve_alf_get_blk_stats -- build container only) and the
goldens tests/golden/make_alf_golden.py records from them."""
import ctypes as C
import os

import numpy as np

from _libs import ORACLE_DIR, ROOT, oracle

GOLDEN = os.path.join(ROOT, "tests", "golden", "alf_v1.npz")
REF_MAIN_SO = os.path.join(ORACLE_DIR, "_ref", "libxevem_ref.so")
M = 3  # MAX_ALF_FILTER_LENGTH >> 1: the margin every kernel may read around its area
BD = 10

# name -> (w, h, content, seed): luma sizes (multiples of 8 as the pictures' are); chroma planes are w/2 x h/2.  Contents: i.i.d. noise (the Laplacian products pass 2^31:
# the reference's int arithmetic wraps there), a smooth drifting texture, stripes in the four directions (every transposition and direction class), flat with a few steps
CASES = {
    "noise_72x72": (72, 72, "noise", 1),
    "binary_noise_136x40": (136, 40, "binary", 2),  # samples 0 / 1023: the Laplacian sums reach 65 000 and `d1 * hv0 > hv1 * d0` wraps
    "texture_96x64": (96, 64, "texture", 3),  # amplitude growing from left to right: every activity level
    "stripes_64x64": (64, 64, "stripes", 4),  # 16x16 tiles of stripes in four orientations, strong and faint: both direction strengths, every transposition
    "steps_40x104": (40, 104, "steps", 5),
    "bright_noise_64x32": (64, 32, "bright", 6),
    # one-sample-period combs at full amplitude: Laplacian sums of 131 000, whose products wrap PAST zero in the reference's int arithmetic -- the comparison that picks
    # the main direction then goes the other way than exact arithmetic would (pinned: the compiled reference decides by the wrapped values)
    "combs_64x48": (64, 48, "combs", 7),
}


class Area(C.Structure):  # AREA (xevem_alf.h:65-71) = xo_alf_area = xeve_hip_alf_area
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


def plane(w, h, content, seed, comp=0):
    """(h + 2M) x (w + 2M) int16 samples, 10 bits; the picture is the interior"""
    r = np.random.default_rng(seed * 10 + comp)
    H, W = h + 2 * M, w + 2 * M
    yy, xx = np.mgrid[0:H, 0:W]
    if content == "noise":
        a = r.integers(0, 1024, size=(H, W))
    elif content == "binary":
        a = 1023 * r.integers(0, 2, size=(H, W))
    elif content == "bright":
        a = 1023 - r.integers(0, 1024, size=(H, W)) // 8 * (r.integers(0, 2, size=(H, W)))
    elif content == "texture":
        amp = (xx / W) ** 2
        a = 512 + amp * (300 * np.sin(xx / 5.0 + yy / 9.0) + 120 * np.cos(yy / 3.0 - xx / 11.0)) + r.integers(-40, 41, size=(H, W)) * amp
    elif content == "stripes":
        q = ((yy // 16) * 4 + xx // 16) % 4  # a 16x16 checker of the four orientations
        s = np.where(q == 0, xx, np.where(q == 1, yy, np.where(q == 2, xx + yy, xx - yy)))
        amp = np.where((yy // 16) % 2 == 0, 400, 12) * np.where((xx // 32) % 2 == 0, 1.0, 0.4)
        a = 512 + amp * np.sin(s * (0.9 + 0.1 * comp)) + r.integers(-3, 4, size=(H, W)) * np.where((xx // 16 + yy // 16) % 3 == 0, 6, 1)
    elif content == "combs":
        q = ((yy // 16) * 4 + xx // 16) % 4
        s = np.where(q == 0, xx, np.where(q == 1, yy, np.where(q == 2, xx + yy, xx - yy)))
        a = np.where(s % 2 == 0, 0, np.where((xx // 32) % 2 == 0, 1023, 700)) + r.integers(0, 3, size=(H, W))
    else:  # steps
        a = 200 + 150 * ((xx // 24) + (yy // 40)) + r.integers(-2, 3, size=(H, W)) * (1 + yy // 32)
    return np.ascontiguousarray(np.clip(a, 0, 1023).astype(np.int16))


def filter_sets(seed):
    """25 x 13 luma coefficients and 7 chroma coefficients as the encoder's quantised filters look (centre near 512 minus twice the rest, 10-bit signed taps) -- and a
    few extreme sets"""
    r = np.random.default_rng(100 + seed)
    luma = r.integers(-40, 41, size=(25, 13)).astype(np.int16)
    luma[:, 12] = (512 - 2 * luma[:, :12].sum(axis=1)).astype(np.int16)
    luma[3] = r.integers(-512, 512, size=13)
    luma[7] = 0
    luma[11, :] = 511
    chroma = r.integers(-60, 61, size=7).astype(np.int16)
    chroma[6] = 512 - 2 * int(chroma[:6].sum())
    return np.ascontiguousarray(luma), np.ascontiguousarray(chroma)


def _p(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


def interior(a):
    """byte pointer of sample (0, 0) of a plane with the margin"""
    return a.ctypes.data + (M * a.shape[1] + M) * a.itemsize


# ---- the oracle -----------------------------------------------------------------------------------------------------------------------------------------------------
def oracle_alf():
    L = oracle()
    L.xo_alf_copy_and_extend.restype = None
    L.xo_alf_copy_and_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.xo_alf_classify.restype = None
    L.xo_alf_classify.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Area), C.c_int]
    L.xo_alf_filter7.restype = None
    L.xo_alf_filter7.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Area), C.c_void_p, C.c_int, C.c_int]
    L.xo_alf_filter5.restype = None
    L.xo_alf_filter5.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Area), C.c_void_p, C.c_int, C.c_int]
    L.xo_alf_blk_stats.restype = None
    L.xo_alf_blk_stats.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class OracleAlf:
    name = "oracle"

    def __init__(self):
        self.L = oracle_alf()

    def classify(self, src, w, h, area):
        """src: plane with margin; area (x, y, w, h) in picture coordinates -> (h, w) uint8 classifier (0 outside the area)"""
        cls = np.zeros((h, w), np.uint8)
        self.L.xo_alf_classify(_p(cls), w, C.c_void_p(interior(src)), src.shape[1], C.byref(Area(*area)), BD)
        return cls

    def filter7(self, cls, src, w, h, area, fset, clip=(0, 1023)):
        dst = np.full((h, w), -1, np.int16)
        x, y, aw, ah = area
        s = src.shape[1]
        self.L.xo_alf_filter7(_p(cls), w, _p(dst, y * w + x), w, C.c_void_p(interior(src) + 2 * (y * s + x)), s, C.byref(Area(*area)), _p(fset), clip[0], clip[1])
        return dst

    def filter5(self, src, w, h, area, fset, clip=(0, 1023)):
        dst = np.full((h, w), -1, np.int16)
        x, y, aw, ah = area
        s = src.shape[1]
        self.L.xo_alf_filter5(_p(dst, y * w + x), w, C.c_void_p(interior(src) + 2 * (y * s + x)), s, C.byref(Area(0, 0, aw, ah)), _p(fset), clip[0], clip[1])
        return dst

    def stats(self, taps, cls, org, rec, w, area):
        """org: (h, w) plane without margin, rec: with margin -> E [nc][13][13], y [nc][13], pix [nc]"""
        nc = 25 if cls is not None else 1
        E, yv, pix = np.zeros((nc, 13, 13)), np.zeros((nc, 13)), np.zeros(nc)
        x, y, aw, ah = area
        self.L.xo_alf_blk_stats(taps, _p(cls) if cls is not None else None, w, _p(org), org.shape[1], C.c_void_p(interior(rec)), rec.shape[1], x, y, aw, ah, _p(E), _p(yv), _p(pix))
        return E, yv, pix

    def copy_and_extend(self, rec, w, h):
        """rec (h, w) -> plane with margin M"""
        tmp = np.full((h + 2 * M, w + 2 * M), -7, np.int16)
        self.L.xo_alf_copy_and_extend(C.c_void_p(interior(tmp)), w + 2 * M, _p(rec), w, w, h, M)
        return tmp


# ---- the reference's own functions (build container) -----------------------------------------------------------------------------------------------------------------
class RefAlf:
    name = "reference"

    def __init__(self):
        L = self.L = C.CDLL(REF_MAIN_SO)
        for f in ("alf_derive_classification_blk", "alf_filter_blk_7", "alf_filter_blk_5", "xeve_alf_get_blk_stats", "alf_init_filter_shape", "alf_cov_create", "alf_copy_and_extend",
                  "alf_copy_and_extend_tile"):
            getattr(L, f).restype = None
        self.shape = {}
        for taps in (5, 7):
            self.shape[taps] = C.create_string_buffer(4096)  # ALF_FILTER_SHAPE (filled by the reference itself)
            L.alf_init_filter_shape(self.shape[taps], C.c_int(taps))

    @staticmethod
    def _rows(a):
        """ALF_CLASSIFIER**: a row-pointer table"""
        rows = (C.c_void_p * a.shape[0])(*[a.ctypes.data + i * a.strides[0] for i in range(a.shape[0])])
        return rows

    def classify(self, src, w, h, area):
        cls = np.zeros((h, w), np.uint8)
        rows = self._rows(cls)
        x0, y0, aw, ah = area
        for i in range(y0, y0 + ah, 32):  # alf_derive_classification (xevem_alf.c:463-486)
            for j in range(x0, x0 + aw, 32):
                a = Area(j, i, min(j + 32, x0 + aw) - j, min(i + 32, y0 + ah) - i)
                self.L.alf_derive_classification_blk(rows, C.c_void_p(interior(src)), C.c_int(src.shape[1]), C.byref(a), C.c_int(BD + 4), C.c_int(BD))
        return cls

    def filter7(self, cls, src, w, h, area, fset, clip=(0, 1023)):
        dst = np.full((h, w), -1, np.int16)
        x, y, aw, ah = area
        s = src.shape[1]
        cr = (C.c_int * 4)(clip[0], clip[1], BD, 0)
        self.L.alf_filter_blk_7(self._rows(cls), _p(dst, y * w + x), C.c_int(w), C.c_void_p(interior(src) + 2 * (y * s + x)), C.c_int(s), C.byref(Area(*area)), C.c_ubyte(0), _p(fset), cr)
        return dst

    def filter5(self, src, w, h, area, fset, clip=(0, 1023)):
        dst = np.full((h, w), -1, np.int16)
        x, y, aw, ah = area
        s = src.shape[1]
        cr = (C.c_int * 4)(clip[0], clip[1], BD, 0)
        self.L.alf_filter_blk_5(None, _p(dst, y * w + x), C.c_int(w), C.c_void_p(interior(src) + 2 * (y * s + x)), C.c_int(s), C.byref(Area(0, 0, aw, ah)), C.c_ubyte(1), _p(fset), cr)
        return dst

    def stats(self, taps, cls, org, rec, w, area):
        class Cov(C.Structure):  # ALF_COVARIANCE (xevem_alf.h:291-297)
            _fields_ = [("num_coef", C.c_int), ("y", C.POINTER(C.c_double)), ("E", C.POINTER(C.POINTER(C.c_double))), ("pix_acc", C.c_double)]

        nc = 25 if cls is not None else 1
        ncoef = taps * taps // 4 + 1
        cov = (Cov * nc)()
        for c in range(nc):
            self.L.alf_cov_create(C.byref(cov[c]), C.c_int(ncoef))
            cov[c].pix_acc = 0.0
        x, y, aw, ah = area
        self.L.xeve_alf_get_blk_stats(C.c_int(0 if cls is not None else 1), cov, self.shape[taps], self._rows(cls) if cls is not None else None, _p(org), C.c_int(org.shape[1]),
                                      C.c_void_p(interior(rec)), C.c_int(rec.shape[1]), C.c_int(x), C.c_int(y), C.c_int(aw), C.c_int(ah))
        E, yv, pix = np.zeros((nc, 13, 13)), np.zeros((nc, 13)), np.zeros(nc)
        for c in range(nc):
            for k in range(ncoef):
                yv[c, k] = cov[c].y[k]
                for l in range(ncoef):
                    E[c, k, l] = cov[c].E[k][l]
            pix[c] = cov[c].pix_acc
        return E, yv, pix

    def copy_and_extend(self, rec, w, h):
        tmp = np.full((h + 2 * M, w + 2 * M), -7, np.int16)
        self.L.alf_copy_and_extend(C.c_void_p(interior(tmp)), C.c_int(w + 2 * M), _p(rec), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(M))
        tile = np.full((h + 2 * M, w + 2 * M), -7, np.int16)
        self.L.alf_copy_and_extend_tile(C.c_void_p(interior(tile)), C.c_int(w + 2 * M), _p(rec), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(M))
        assert np.array_equal(tmp, tile)  # (the two forms differ in nothing but a literal 3 for the margin)
        return tmp


def run_case(impl, name):
    """every kernel of one case -> dict of arrays (what the golden file holds per case)"""
    w, h, content, seed = CASES[name]
    luma, cb = plane(w, h, content, seed, 0), plane(w // 2, h // 2, content, seed, 1)
    org_l = np.ascontiguousarray(plane(w, h, content, seed + 50, 0)[M:M + h, M:M + w])
    org_c = np.ascontiguousarray(plane(w // 2, h // 2, content, seed + 50, 1)[M:M + h // 2, M:M + w // 2])
    fl, fc = filter_sets(seed)
    out = {}
    whole = (0, 0, w, h)
    cls = impl.classify(luma, w, h, whole)
    out["cls"] = cls
    # a CTU-shaped piece away from the origin (its classifier entries must equal the whole picture's: the class of a 4x4 block depends on its 10x10 window alone)
    piece = (8, 4, min(64, w - 8) // 4 * 4, min(64, h - 4) // 4 * 4)
    out["cls_piece"] = impl.classify(luma, w, h, piece)
    out["f7"] = impl.filter7(cls, luma, w, h, whole, fl)
    out["f7_piece_clip"] = impl.filter7(cls, luma, w, h, piece, fl, clip=(64, 940))
    out["f5"] = impl.filter5(cb, w // 2, h // 2, (0, 0, w // 2, h // 2), fc)
    out["f5_piece"] = impl.filter5(cb, w // 2, h // 2, (4, 8, (w // 2 - 4) // 4 * 4, (h // 2 - 8) // 4 * 4), fc, clip=(16, 1000))
    for taps in (5, 7):
        E, yv, pix = impl.stats(taps, cls, org_l, luma, w, whole)
        out["E%d" % taps], out["y%d" % taps], out["pix%d" % taps] = E, yv, pix
    E, yv, pix = impl.stats(7, cls, org_l, luma, w, piece)
    out["E7_piece"], out["y7_piece"], out["pix7_piece"] = E, yv, pix
    E, yv, pix = impl.stats(5, None, org_c, cb, w // 2, (0, 0, w // 2, h // 2))
    out["Ec"], out["yc"], out["pixc"] = E, yv, pix
    out["ext"] = impl.copy_and_extend(org_l, w, h)
    return out


# ---- the kernels' per-element code on the host (tests/native/alf_host.cpp over xeve_amd/csrc/alf_core.h) ---------------------------------------------------------------
HOST_SRC = os.path.join(ROOT, "tests", "native", "alf_host.cpp")
HOST_HDR = os.path.join(ROOT, "xeve_amd", "csrc", "alf_core.h")
HOST_OUT = os.path.join(ROOT, "tests", "native", "build", "libalf_host.so")


class HostAlf(OracleAlf):
    """alf_core.h compiled by g++: classification, filters and statistics from the kernels' own per-lane functions; copy_and_extend stays the oracle's (a clamp, in alf.hip)"""
    name = "alf_core.h on the host"

    def __init__(self):
        import subprocess

        OracleAlf.__init__(self)
        if not os.path.exists(HOST_OUT) or os.path.getmtime(HOST_OUT) < max(os.path.getmtime(HOST_SRC), os.path.getmtime(HOST_HDR)):
            os.makedirs(os.path.dirname(HOST_OUT), exist_ok=True)
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_OUT, HOST_SRC], check=True)
        self.H = C.CDLL(HOST_OUT)
        for f in ("xa_host_classify", "xa_host_filter", "xa_host_stats"):
            getattr(self.H, f).restype = None

    def classify(self, src, w, h, area):
        cls = np.zeros((h, w), np.uint8)
        self.H.xa_host_classify(_p(cls), C.c_int(w), C.c_void_p(interior(src)), C.c_int(src.shape[1]), *[C.c_int(v) for v in area], C.c_int(BD))
        return cls

    def _filter(self, taps, cls, src, w, h, area, fset, clip):
        dst = np.full((h, w), -1, np.int16)
        x, y, aw, ah = area
        s = src.shape[1]
        self.H.xa_host_filter(C.c_int(taps), _p(cls) if cls is not None else None, C.c_int(w), _p(dst, y * w + x), C.c_int(w), C.c_void_p(interior(src) + 2 * (y * s + x)), C.c_int(s),
                              C.c_int(x), C.c_int(y), C.c_int(aw), C.c_int(ah), _p(fset), C.c_int(clip[0]), C.c_int(clip[1]))
        return dst

    def filter7(self, cls, src, w, h, area, fset, clip=(0, 1023)):
        return self._filter(7, cls, src, w, h, area, fset, clip)

    def filter5(self, src, w, h, area, fset, clip=(0, 1023)):
        return self._filter(5, None, src, w, h, area, fset, clip)

    def stats(self, taps, cls, org, rec, w, area):
        nc = 25 if cls is not None else 1
        E, yv, pix = np.full((nc, 13, 13), -1.0), np.full((nc, 13), -1.0), np.full(nc, -1.0)
        self.H.xa_host_stats(C.c_int(taps), _p(cls) if cls is not None else None, C.c_int(w), _p(org), C.c_int(org.shape[1]), C.c_void_p(interior(rec)), C.c_int(rec.shape[1]),
                             *[C.c_int(v) for v in area], _p(E), _p(yv), _p(pix))
        return E, yv, pix
