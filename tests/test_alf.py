"""Main profile, the adaptive loop filter's sample kernels ("ALF classification / filter"): alf_derive_classification, alf_filter_blk_7 / _5,
xeve_alf_get_blk_stats + xeve_alf_clac_covariance, alf_copy_and_extend (src_main/xevem_alf.c).
  (cpu) the oracle's restatement against the goldens recorded from the reference's own functions, and against those functions called in place where oracle/_ref exists;
  (cpu) the kernels' per-element code (xeve_amd/csrc/alf_core.h, __host__ __device__) compiled for the host against the oracle;
  (gpu) the HIP entry points (xeve_hip_alf_*) against oracle and goldens."""
import os

import numpy as np
import pytest

import _alf

GOLD = np.load(_alf.GOLDEN)


def _check(impl, name, against):
    got = _alf.run_case(impl, name)
    for k, v in got.items():
        want = against(name, k)
        assert v.shape == want.shape and v.dtype == want.dtype and np.array_equal(v, want), (impl.name, name, k)
    return got


def golden(name, k):
    return GOLD[name + "/" + k]


@pytest.mark.parametrize("name", sorted(_alf.CASES))
def test_oracle_alf_matches_the_reference_goldens(name):
    got = _check(_alf.OracleAlf(), name, golden)
    # the class of a 4x4 block is a function of its 10x10 window alone: a piece classified on its own = the same entries of the whole picture
    x, y, w, h = 8, 4, got["cls_piece"].shape[1], got["cls_piece"].shape[0]
    piece = got["cls_piece"]
    ys, xs = np.nonzero(piece)
    assert ys.size and np.array_equal(piece[ys.min():ys.max() + 1, xs.min():xs.max() + 1], got["cls"][ys.min():ys.max() + 1, xs.min():xs.max() + 1])


def test_the_cases_reach_most_classes_and_every_transposition():
    classes, trans = set(), set()
    for name in _alf.CASES:
        c = GOLD[name + "/cls"]
        classes |= set(np.unique(c >> 2).tolist())
        trans |= set(np.unique(c & 3).tolist())
    assert len(classes) >= 16 and trans == {0, 1, 2, 3}
    # every activity level 0 .. 4 without a direction, and both direction strengths in both parities of the main direction (offsets 5, 10, 15, 20)
    assert {0, 1, 2, 3, 4} <= classes and all(any(5 * k <= c < 5 * k + 5 for c in classes) for k in (1, 2, 3, 4))


@pytest.mark.ref
@pytest.mark.skipif(not os.path.exists(_alf.REF_MAIN_SO), reason="oracle/_ref (Main profile) not built")
@pytest.mark.parametrize("name", sorted(_alf.CASES))
def test_oracle_alf_matches_the_reference_in_place(name):
    ref = _alf.run_case(_alf.RefAlf(), name)
    _check(_alf.OracleAlf(), name, lambda n, k: ref[k])
    for k, v in ref.items():  # (and the committed goldens are what the reference produces today)
        assert np.array_equal(v, GOLD[name + "/" + k]), k


def test_statistics_add_up_over_pieces():
    """the per-CTU records add up to the picture's (xeve_alf_get_frame_stat, xevem_alf.c:3729-3743): exact integers in doubles, whatever the order"""
    O = _alf.OracleAlf()
    w, h, content, seed = _alf.CASES["texture_96x64"]
    luma = _alf.plane(w, h, content, seed, 0)
    org = np.ascontiguousarray(_alf.plane(w, h, content, seed + 50, 0)[_alf.M:_alf.M + h, _alf.M:_alf.M + w])
    cls = O.classify(luma, w, h, (0, 0, w, h))
    E, yv, pix = O.stats(7, cls, org, luma, w, (0, 0, w, h))
    Es, ys, ps = np.zeros_like(E), np.zeros_like(yv), np.zeros_like(pix)
    for y0 in range(0, h, 32):
        for x0 in range(0, w, 32):
            e, y_, p_ = O.stats(7, cls, org, luma, w, (x0, y0, 32, 32))
            Es += e
            ys += y_
            ps += p_
    assert np.array_equal(E, Es) and np.array_equal(yv, ys) and np.array_equal(pix, ps)
    assert np.array_equal(E, np.transpose(E, (0, 2, 1)))


@pytest.mark.parametrize("name", sorted(_alf.CASES))
def test_the_kernels_per_lane_code_on_the_host_matches_the_goldens(name):
    """xeve_amd/csrc/alf_core.h (block_class, filter_sample, local_sums, stat_entry: what a lane of alf.hip runs) compiled for the host, driven in the kernels' decomposition"""
    _check(_alf.HostAlf(), name, golden)


# ---- the HIP entry points ------------------------------------------------------------------------------------------------------------------------------------------------
FILTER_JOB = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("dst_off", "<i8"), ("src_off", "<i8")])  # xeve_hip_alf_filter_job
AREA = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])  # xeve_hip_alf_area
assert FILTER_JOB.itemsize == 32 and AREA.itemsize == 16


class HipAlf(_alf.OracleAlf):
    """xeve_hip_alf_* on planes resident in HBM (torch tensors as device memory); an area larger than 64x64 is handed over as ONE job (the kernel tiles it) and, for the
    statistics, also as 64x64 jobs whose records are added up"""
    name = "hip"

    def __init__(self):
        import torch

        import xeve_amd
        from xeve_amd import lib

        xeve_amd.init(0)
        self.t, self.L, self.check = torch, lib.load(), lib.check
        self.dev = torch.device("cuda:0")

    def up(self, a):
        return self.t.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(self.dev)

    def classify(self, src, w, h, area):
        d_src, cls = self.up(src), self.t.zeros(h * w, dtype=self.t.uint8, device=self.dev)
        a = _alf.Area(*area)
        self.check(self.L.xeve_hip_alf_classify(cls.data_ptr(), w, d_src.data_ptr() + 2 * (_alf.M * src.shape[1] + _alf.M), src.shape[1], _alf.C.addressof(a), _alf.BD, None))
        self.t.cuda.synchronize()
        return cls.cpu().numpy().reshape(h, w)

    def _filter(self, taps, cls, src, w, h, area, fset, clip):
        x, y, aw, ah = area
        s = src.shape[1]
        d_src, dst = self.up(src), self.t.full((h * w,), -1, dtype=self.t.int16, device=self.dev)
        d_cls = self.up(cls) if cls is not None else None
        job = np.zeros(1, FILTER_JOB)
        job[0] = (x, y, aw, ah, y * w + x, (_alf.M + y) * s + _alf.M + x)
        d_job = self.up(job)
        self.check(self.L.xeve_hip_alf_filter_jobs(taps, dst.data_ptr(), w, d_src.data_ptr(), s, d_cls.data_ptr() if d_cls is not None else None, w, d_job.data_ptr(), 1,
                                                   fset.ctypes.data, clip[0], clip[1], None))
        self.t.cuda.synchronize()
        return dst.cpu().numpy().reshape(h, w)

    def filter7(self, cls, src, w, h, area, fset, clip=(0, 1023)):
        return self._filter(7, cls, src, w, h, area, fset, clip)

    def filter5(self, src, w, h, area, fset, clip=(0, 1023)):
        return self._filter(5, None, src, w, h, area, fset, clip)

    def stats(self, taps, cls, org, rec, w, area):
        nc = 25 if cls is not None else 1
        x, y, aw, ah = area
        out = []
        for jobs in ([(x, y, aw, ah)], [(x + i, y + j, min(64, aw - i), min(64, ah - j)) for j in range(0, ah, 64) for i in range(0, aw, 64)]):
            ja = np.zeros(len(jobs), AREA)
            for i, jb in enumerate(jobs):
                ja[i] = jb
            n = len(jobs)
            E, yv, pix = (self.t.full((n * nc * k,), -1.0, dtype=self.t.float64, device=self.dev) for k in (169, 13, 1))
            d_org, d_rec, d_job = self.up(org), self.up(rec), self.up(ja)
            d_cls = self.up(cls) if cls is not None else None
            self.check(self.L.xeve_hip_alf_blk_stats_jobs(taps, d_cls.data_ptr() if d_cls is not None else None, w, d_org.data_ptr(), org.shape[1],
                                                          d_rec.data_ptr() + 2 * (_alf.M * rec.shape[1] + _alf.M), rec.shape[1], d_job.data_ptr(), n, E.data_ptr(), yv.data_ptr(),
                                                          pix.data_ptr(), None))
            self.t.cuda.synchronize()
            out.append((E.cpu().numpy().reshape(n, nc, 13, 13).sum(axis=0), yv.cpu().numpy().reshape(n, nc, 13).sum(axis=0), pix.cpu().numpy().reshape(n, nc).sum(axis=0)))
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)  # (one job tiled by the kernel = its 64x64 jobs added up)
        return out[0]

    def copy_and_extend(self, rec, w, h):
        m = _alf.M
        d_rec, tmp = self.up(rec), self.t.full(((h + 2 * m) * (w + 2 * m),), -7, dtype=self.t.int16, device=self.dev)
        self.check(self.L.xeve_hip_alf_copy_and_extend(tmp.data_ptr() + 2 * (m * (w + 2 * m) + m), w + 2 * m, d_rec.data_ptr(), w, w, h, m, None))
        self.t.cuda.synchronize()
        return tmp.cpu().numpy().reshape(h + 2 * m, w + 2 * m)


class HipAlfHost(_alf.RefAlf):
    """the host-memory forms with the reference's signatures (what alf->derive_classification_blk / filter_7x7_blk / filter_5x5_blk can be pointed at): driven exactly as
    tests/_alf.py drives the reference's own functions"""
    name = "hip host forms"

    def __init__(self):
        import xeve_amd
        from xeve_amd import lib

        xeve_amd.init(0)
        L = lib.load()

        class Fns:
            alf_derive_classification_blk = L.xeve_hip_alf_derive_classification_blk_host
            alf_filter_blk_7 = L.xeve_hip_alf_filter_blk_7_host
            alf_filter_blk_5 = L.xeve_hip_alf_filter_blk_5_host

        self.L, self.hip = Fns, L

    def stats(self, taps, cls, org, rec, w, area, preset=0.0):
        """xeve_hip_alf_get_blk_stats_host on records that already hold `preset` in their upper triangles: added to, the lower triangles set from the upper ones"""
        C = _alf.C

        class Cov(C.Structure):  # xeve_hip_alf_covariance = ALF_COVARIANCE
            _fields_ = [("num_coef", C.c_int), ("y", C.POINTER(C.c_double)), ("E", C.POINTER(C.POINTER(C.c_double))), ("pix_acc", C.c_double)]

        nc, ncoef = (25 if cls is not None else 1), taps * taps // 4 + 1
        Es, ys = np.full((nc, ncoef, ncoef), preset), np.full((nc, ncoef), preset)
        rows = [(C.POINTER(C.c_double) * ncoef)(*[C.cast(Es[c, k].ctypes.data, C.POINTER(C.c_double)) for k in range(ncoef)]) for c in range(nc)]
        cov = (Cov * nc)()
        for c in range(nc):
            cov[c].num_coef, cov[c].y, cov[c].E, cov[c].pix_acc = ncoef, C.cast(ys[c].ctypes.data, C.POINTER(C.c_double)), rows[c], preset
        x, y, aw, ah = area
        self.hip.xeve_hip_alf_get_blk_stats_host(taps, cov, self._rows(cls) if cls is not None else None, org.ctypes.data, org.shape[1], _alf.interior(rec), rec.shape[1], x, y, aw, ah)
        E, yv, pix = np.zeros((nc, 13, 13)), np.zeros((nc, 13)), np.zeros(nc)
        E[:, :ncoef, :ncoef], yv[:, :ncoef] = Es, ys
        for c in range(nc):
            pix[c] = cov[c].pix_acc
        return E, yv, pix

    def copy_and_extend(self, *a):
        raise NotImplementedError


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_alf.CASES))
def test_hip_alf_matches_oracle_and_goldens(name):
    want = _alf.run_case(_alf.OracleAlf(), name)
    got = _check(HipAlf(), name, golden)
    assert all(np.array_equal(got[k], want[k]) for k in want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stripes_64x64", "combs_64x48", "noise_72x72"])
def test_hip_alf_host_forms_with_the_references_signatures(name):
    w, h, content, seed = _alf.CASES[name]
    H = HipAlfHost()
    luma, cb = _alf.plane(w, h, content, seed, 0), _alf.plane(w // 2, h // 2, content, seed, 1)
    fl, fc = _alf.filter_sets(seed)
    cls = H.classify(luma, w, h, (0, 0, w, h))
    assert np.array_equal(cls, GOLD[name + "/cls"])
    assert np.array_equal(H.filter7(cls, luma, w, h, (0, 0, w, h), fl), GOLD[name + "/f7"])
    piece = (8, 4, min(64, w - 8) // 4 * 4, min(64, h - 4) // 4 * 4)
    assert np.array_equal(H.filter7(cls, luma, w, h, piece, fl, clip=(64, 940)), GOLD[name + "/f7_piece_clip"])
    assert np.array_equal(H.filter5(cb, w // 2, h // 2, (0, 0, w // 2, h // 2), fc), GOLD[name + "/f5"])
    # the statistics form adds into the caller's records (upper triangle, y, energy) and mirrors the triangle, as xeve_alf_get_blk_stats does
    org = np.ascontiguousarray(_alf.plane(w, h, content, seed + 50, 0)[_alf.M:_alf.M + h, _alf.M:_alf.M + w])
    for taps in (5, 7):
        E, yv, pix = H.stats(taps, cls, org, luma, w, (0, 0, w, h))
        assert np.array_equal(E, GOLD[name + "/E%d" % taps]) and np.array_equal(yv, GOLD[name + "/y%d" % taps]) and np.array_equal(pix, GOLD[name + "/pix%d" % taps])
    E, yv, pix = H.stats(7, cls, org, luma, w, piece, preset=1000.0)
    n = 13
    want = GOLD[name + "/E7_piece"] + 1000.0
    assert np.array_equal(E[:, :n, :n], want) and np.array_equal(yv, GOLD[name + "/y7_piece"] + 1000.0) and np.array_equal(pix, GOLD[name + "/pix7_piece"] + 1000.0)
    org_c = np.ascontiguousarray(_alf.plane(w // 2, h // 2, content, seed + 50, 1)[_alf.M:_alf.M + h // 2, _alf.M:_alf.M + w // 2])
    E, yv, pix = H.stats(5, None, org_c, cb, w // 2, (0, 0, w // 2, h // 2))
    assert np.array_equal(E, GOLD[name + "/Ec"]) and np.array_equal(yv, GOLD[name + "/yc"]) and np.array_equal(pix, GOLD[name + "/pixc"])
