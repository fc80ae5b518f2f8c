#!/usr/bin/env python3
"""Times the Main-profile sample kernels of round 5 on one picture resident in HBM (HIP events around repeated launches on the default stream) and prices them against the
HBM roof: the adaptive loop filter's copy-and-extend, classification, 7x7 / 5x5 filters and statistics (one job per CTU), and affine motion compensation (one CU per
workgroup).  Algorithmic bytes = what the kernel must read and write once.  usage: probe_main_kernels.py [--width W --height H --reps N] > out.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch

    import xeve_amd
    from xeve_amd import lib

    xeve_amd.init(0)
    L, dev = lib.load(), torch.device("cuda:0")
    W, H, M, R = a.width, a.height, 3, a.reps
    out = {"picture": [W, H], "reps": R, "hbm_peak_GBs": 8000.0, "kernels": {}}

    def timed(name, alg_bytes, fn):
        try:
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(R):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / R
            out["kernels"][name] = {"ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes), "GBs": round(alg_bytes / ms / 1e6, 1), "frac_of_hbm_peak": round(alg_bytes / ms / 1e6 / 8000.0, 4)}
        except Exception as e:  # noqa: BLE001 -- one kernel's failure must not lose the others' figures
            out["kernels"][name] = {"error": repr(e)[:300]}

    def check(rc):
        lib.check(rc)

    g = torch.Generator(device=dev)
    g.manual_seed(3)
    S = W + 2 * M
    rec = torch.randint(0, 1024, (H * W,), dtype=torch.int16, device=dev, generator=g)
    org = torch.randint(0, 1024, (H * W,), dtype=torch.int16, device=dev, generator=g)
    ext = torch.zeros(((H + 2 * M) * S,), dtype=torch.int16, device=dev)
    dst = torch.zeros((H * W,), dtype=torch.int16, device=dev)
    cls = torch.zeros((H * W,), dtype=torch.uint8, device=dev)
    ext0 = ext.data_ptr() + 2 * (M * S + M)
    timed("alf_copy_and_extend", 4 * W * H, lambda: check(L.xeve_hip_alf_copy_and_extend(ext0, S, rec.data_ptr(), W, W, H, M, None)))
    area = np.array([(0, 0, W, H)], dtype=[("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
    timed("alf_classify", 3 * W * H, lambda: check(L.xeve_hip_alf_classify(cls.data_ptr(), W, ext0, S, area.ctypes.data, 10, None)))
    # one job per CTU, as xeve_alf_recon / xeve_alf_derive_stats_filtering walk the picture
    ctus = [(x, y, min(64, W - x), min(64, H - y)) for y in range(0, H, 64) for x in range(0, W, 64)]
    fj = np.zeros(len(ctus), dtype=[("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("dst_off", "<i8"), ("src_off", "<i8")])
    sj = np.zeros(len(ctus), dtype=[("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
    for i, (x, y, w, h) in enumerate(ctus):
        fj[i] = (x, y, w, h, y * W + x, (M + y) * S + M + x)
        sj[i] = (x, y, w, h)
    d_fj = torch.from_numpy(fj.view(np.uint8).reshape(-1).copy()).to(dev)
    d_sj = torch.from_numpy(sj.view(np.uint8).reshape(-1).copy()).to(dev)
    r = np.random.default_rng(1)
    f7 = np.ascontiguousarray(r.integers(-40, 41, size=(25, 13)).astype(np.int16))
    f7[:, 12] = 512 - 2 * f7[:, :12].sum(axis=1)
    f5 = np.ascontiguousarray(np.array([3, -7, 12, -7, 9, 20, 452], np.int16))
    n = len(ctus)
    timed("alf_filter_7x7", (4 + 1 / 16) * W * H, lambda: check(L.xeve_hip_alf_filter_jobs(7, dst.data_ptr(), W, ext.data_ptr(), S, cls.data_ptr(), W, d_fj.data_ptr(), n, f7.ctypes.data, 0, 1023, None)))
    timed("alf_filter_5x5 (one plane of this size)", 4 * W * H, lambda: check(L.xeve_hip_alf_filter_jobs(5, dst.data_ptr(), W, ext.data_ptr(), S, None, W, d_fj.data_ptr(), n, f5.ctypes.data, 0, 1023, None)))
    E = torch.zeros((n * 25 * 169,), dtype=torch.float64, device=dev)
    yv = torch.zeros((n * 25 * 13,), dtype=torch.float64, device=dev)
    px = torch.zeros((n * 25,), dtype=torch.float64, device=dev)
    for taps in (7, 5):
        timed("alf_blk_stats_%d (luma, 25 classes)" % taps, (4 + 1 / 16) * W * H + n * 25 * 183 * 8,
              lambda taps=taps: check(L.xeve_hip_alf_blk_stats_jobs(taps, cls.data_ptr(), W, org.data_ptr(), W, ext0, S, d_sj.data_ptr(), n, E.data_ptr(), yv.data_ptr(), px.data_ptr(), None)))
    # affine MC: every 32x32 CU of the picture, bi-prediction, a rotation-like model (sub-blocks) and a zoom (enhanced filter); planes padded by 160 / 80
    P = 160
    SL, SC = W + 2 * P, W // 2 + P
    planes = [torch.randint(0, 1024, ((H + 2 * P) * SL,), dtype=torch.int16, device=dev, generator=g)] + [torch.randint(0, 1024, ((H // 2 + P) * SC,), dtype=torch.int16, device=dev, generator=g) for _ in range(2)]
    tab = np.zeros(2, dtype=[("y", "<u8"), ("u", "<u8"), ("v", "<u8"), ("poc", "<i4"), ("pad_", "<i4")])
    for l in range(2):
        tab[l] = (planes[0].data_ptr() + 2 * (P * SL + P), planes[1].data_ptr() + 2 * (P // 2 * SC + P // 2), planes[2].data_ptr() + 2 * (P // 2 * SC + P // 2), l, 0)
    JOB = np.dtype([("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 3, 2)), ("refi", "i1", (2,)), ("vertex_num", "i1"), ("pad_", "i1")])
    cw = 32
    cus = [(x, y) for y in range(0, H - cw + 1, cw) for x in range(0, W - cw + 1, cw)]
    for label, d1, d2 in (("sub-block filters", (1, 0), (0, 1)), ("enhanced interpolation filter", (40, 0), (0, 0))):
        jobs = np.zeros(len(cus), JOB)
        for i, (x, y) in enumerate(cus):
            jobs[i]["x"], jobs[i]["y"], jobs[i]["refi"], jobs[i]["vertex_num"] = x, y, (0, 0), 3
            for l in range(2):
                base = np.array([5 + l * 3 + (i % 7), -9 + (i % 5)])
                jobs[i]["mv"][l] = [base, base + d1, base + d2]
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
        py, pu, pv = (torch.zeros((len(cus) * cw * cw // k,), dtype=torch.int16, device=dev) for k in (1, 4, 4))
        # per CU and list: the luma and two chroma windows once (+ filter margins), the prediction written once (and read back for the second list)
        alg = len(cus) * (2 * ((cw + 7) ** 2 + 2 * (cw // 2 + 3) ** 2) * 2 + 3 * cw * cw * 3 // 2 * 2)
        timed("affine_mc 32x32 bi (%s)" % label, alg,
              lambda d_jobs=d_jobs, py=py, pu=pu, pv=pv: check(L.xeve_hip_affine_mc_jobs(tab.ctypes.data, 1, 1, SL, SC, W, H, d_jobs.data_ptr(), len(cus), cw, cw, 10, py.data_ptr(), pu.data_ptr(), pv.data_ptr(), None)))
        out["kernels"]["affine_mc 32x32 bi (%s)" % label]["cus"] = len(cus)
    # the affine gradient search: every 32x32 (64x64) CU of the picture in ONE launch; a smooth texture, the original = the reference moved by a fraction of a sample + noise,
    # so the searches run their rounds; algorithmic bytes at the full round budget: per compensation the window + per SATD and per error pass the original
    yy, xx = torch.meshgrid(torch.arange(H + 2 * P, device=dev, dtype=torch.float32), torch.arange(SL, device=dev, dtype=torch.float32), indexing="ij")
    tex = lambda x, y: 512 + 280 * torch.sin(x / 6.0 + y / 9.0) + 150 * torch.cos(y / 4.0 - x / 13.0)  # noqa: E731
    ref = (tex(xx, yy) + torch.randint(-20, 21, xx.shape, device=dev, generator=g)).clamp(0, 1023).to(torch.int16).reshape(-1).contiguous()
    cur = (tex(xx * 1.004 + 1.3, yy * 0.997 - 0.8) + torch.randint(-20, 21, xx.shape, device=dev, generator=g)).clamp(0, 1023).to(torch.int16)[P:P + H, P:P + W].contiguous()
    tab_me = np.zeros(2, dtype=tab.dtype)
    for l in range(2):
        tab_me[l] = (ref.data_ptr() + 2 * (P * SL + P), 0, 0, l, 0)
    MEJOB = np.dtype([("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (3, 2)), ("mv", "<i2", (3, 2)), ("refi", "i1"), ("list", "i1"), ("bi", "i1"), ("vertex_num", "i1"), ("mot_bits_other", "<i4"),
                      ("cost", "<u4")])
    for cw, vn in ((32, 2), (32, 3), (64, 2), (64, 3)):
        cus = [(x, y) for y in range(0, H - cw + 1, cw) for x in range(0, W - cw + 1, cw)]
        jobs = np.zeros(len(cus), MEJOB)
        for i, (x, y) in enumerate(cus):
            jobs[i]["x"], jobs[i]["y"], jobs[i]["vertex_num"] = x, y, vn
        h_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy())
        d_jobs = h_jobs.to(dev)
        rounds = 7 - (2 if vn == 3 else 0)
        alg = len(cus) * ((rounds + 1) * ((cw + 7) ** 2 + cw * cw) * 2 + rounds * cw * cw * 2 + 88)

        def run(d_jobs=d_jobs, h_jobs=h_jobs, cw=cw, n=len(cus)):
            d_jobs.copy_(h_jobs, non_blocking=True)  # (the launch moves the jobs' vectors: every repetition starts from the same ones)
            check(L.xeve_hip_affine_me_jobs(tab_me.ctypes.data, 1, 1, SL, W, H, cur.data_ptr(), W, None, d_jobs.data_ptr(), n, cw, cw, 10, 1234567, 1, 1, None))

        name = "affine_me %dx%d uni, %d control points" % (cw, cw, vn)
        timed(name, alg, run)
        res = d_jobs.cpu().numpy().view(MEJOB)
        out["kernels"][name].update({"searches": len(cus), "moved": int(np.sum(np.any(res["mv"].reshape(len(cus), -1) != 0, axis=1))),
                                     "searches_per_s": round(len(cus) / out["kernels"][name]["ms"] * 1e3) if "ms" in out["kernels"][name] else None})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
