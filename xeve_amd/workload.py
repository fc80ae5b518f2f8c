"""The hot-path pass over ONE inter picture: the unit of work bench.py times and tests/test_workload.py checks.

What the reference does per inter picture on this path -- for every quad-tree
level S in {8,16,32,64} and every block of that size (mode_coding_tree visits all levels, xeve_mode.c:2007):

  A. integer motion search, per reference list (2 lists): `me_ipel_diamond` rounds (xeve_pinter.c:363-551) --
     a dense 5x5 grid plus 4/8/16-point diamonds of growing radius around a centre, ~80-100 SAD calls per
     pass, repeated while the best moves (xeve_pinter.c:784-822).  Modelled as N_PASS = 3 passes of the
     90-candidate pattern per list => 540 SAD calls per block, matching the probed 533..619 calls per block
     per picture.
  B. half-pel refinement, per list: 4 positions (preset medium, xeve_enc.c:2464), each = one luma
     interpolation (n0, nn, 0n, nn) + one SAD against the dense prediction (xeve_pinter.c:593-627).
  C. skip/merge analysis: 3 candidates (merge_num 3) x [MC Y,U,V + SSD Y,U,V] (xeve_pinter.c:1337-1458).
  D. inter residual RDO of the winner (pinter_residue_rdo, xeve_pinter.c:906-1051): bi-predicted MC (two
     lists + average) for Y,U,V; DIFF; forward transform; quantisation (RDOQ zero pre-test + plain quant);
     dequantisation; inverse transform; reconstruction; SSD of prediction and of reconstruction.
  E. intra gate: SATD of the block against the prediction (xeve_mode.c:1252).

Everything runs through the batched C-ABI on planes resident in HBM.  Motion vectors are synthetic (uniform
in +-MV_RANGE) because the sequential RDO that would choose them is outside this tier's scope; the amount and
shape of arithmetic per picture is the reference's.
"""
import numpy as np
import torch

from . import device as D

PAD_L, PAD_C = 144, 72  # picture plane padding (reference: src_base/xeve_def.h:380-381)
SIZES = (8, 16, 32, 64)
N_LIST, N_PASS = 2, 3
HALF_PEL = ((-8, 0), (-8, 8), (0, 8), (8, 8))  # 1/16-pel offsets of the 4 medium-preset half-pel points (xeve_pinter.c:67-70)
N_MERGE = 3
import os
SORT_JOBS = os.environ.get("XEVE_SORT_JOBS", "1") == "1"
USE_RDOQ = os.environ.get("XEVE_RDOQ", "1") == "1"  # developer switch: 0 = plain quantiser in phase D
MV_RANGE = 48  # integer-pel; keeps centre +- 64 diamond inside the 144-pel padding


def diamond_pattern():
    """(dx, dy) of one me_ipel_diamond pass: 5x5 dense, then steps 4 (4 pts), 8 (8 pts), 16/32/64 (16 pts), each
    followed by a re-test of the centre (xeve_pinter.c:405-540)."""
    c = [(dx, dy) for dy in range(-2, 3) for dx in range(-2, 3)]
    for st in (4, 8, 16, 32, 64):
        n = 4 if st == 4 else (8 if st == 8 else 16)
        q = n // 4
        for i in range(n):
            a, b = i % q, i // q
            dx, dy = [(a, -(q - a)), (q - a, a), (-a, q - a), (-(q - a), -a)][b]
            c.append((dx * st // q, dy * st // q))
        c.append((0, 0))
    return c


class HotPathPass:
    def __init__(self, width, height, device, seed=4, bit_depth=10, qp=32, sizes=SIZES, content="iid"):
        assert width % 64 == 0 and height % 8 == 0
        self.W, self.H, self.dev, self.bd, self.qp, self.sizes = width, height, device, bit_depth, qp, tuple(sizes)
        self.s_l, self.s_c = width + 2 * PAD_L, width // 2 + 2 * PAD_C
        g = torch.Generator(device=device).manual_seed(seed)
        hl, hc = height + 2 * PAD_L, height // 2 + 2 * PAD_C
        self.content = content
        self.level_threads = os.environ.get("XEVE_HIP_LEVEL_THREADS", "0") == "1"
        if content == "iid":
            # synthetic i.i.d. uniform picture planes: the original is an 8-bit source << (bit_depth - 8), as the encoder sees the BASELINE configs' random
            # 8-bit YUV (xeve_app converts on input); the reference pictures are reconstructions, any value of the internal depth
            mk = lambda h, s: torch.randint(0, 1 << bit_depth, (h, s), generator=g, device=device, dtype=torch.int16)
            mk8 = lambda h, s: (torch.randint(0, 256, (h, s), generator=g, device=device, dtype=torch.int16) << (bit_depth - 8))
            self.org = [mk8(hl, self.s_l), mk8(hc, self.s_c), mk8(hc, self.s_c)]
            self.ref = [[mk(hl, self.s_l), mk(hc, self.s_c), mk(hc, self.s_c)] for _ in range(N_LIST)]
        else:
            # SURVEY.md 8(d)'s structured input: a moving gradient + 3-bit noise, ((x + 3f) * 2 + (y + f) + rand3) & 255 as an
            # 8-bit source (<< 2).  The current picture is frame 1, list 0 holds frame 0, list 1 frame 2, so the true motion
            # is (+3, +1) / (-3, -1) luma samples and phase D predicts as well as real video does.
            def frame(f, h, s, pad, sub):
                y = (torch.arange(h, device=device) - pad)[:, None] * sub
                x = (torch.arange(s, device=device) - pad)[None, :] * sub
                n = torch.randint(0, 8, (h, s), generator=g, device=device)
                return ((((x + 3 * f) * 2 + (y + f) + n) & 255) << (bit_depth - 8)).to(torch.int16)
            planes = lambda f: [frame(f, hl, self.s_l, PAD_L, 1), frame(f, hc, self.s_c, PAD_C, 2), frame(f, hc, self.s_c, PAD_C, 2)]
            self.org, self.ref = planes(1), [planes(0), planes(2)]
        # alignment copies of the luma reference planes (xeve_hip_sad_jobs_dual): made once per reference picture
        self.ref_s1 = [D.plane_shift1(r[0]) for r in self.ref]
        self.pattern = diamond_pattern()
        # RDOQ inputs: lambda of the reference for this qp (xeve_enc.c: lambda = 0.57 * 2^((qp - 12) / 3)) and bit estimates of
        # equiprobable context models (entropy_bits of state 256, xeve_mode.c:304-325) -- the values a fresh CABAC state gives
        self.lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
        from .lib import RdoqEst
        e = RdoqEst()
        eq = 32768  # -32768 * (log2(256.5 / 512) - 9 + 9) ~ one bit
        e.cbf[0] = e.cbf[1] = eq
        for i in range(24):
            e.run[i][0] = e.run[i][1] = e.level[i][0] = e.level[i][1] = eq
        for i in range(2):
            e.last[i][0] = e.last[i][1] = eq
        self.rdoq_est = e
        self.cand_l = torch.tensor([dy * self.s_l + dx for dx, dy in self.pattern], dtype=torch.int32, device=device)
        self.zero_cand = torch.zeros(1, dtype=torch.int32, device=device)
        rng = np.random.default_rng(seed)
        self.lv = {}
        for S in self.sizes:
            self.lv[S] = self._level(S, rng)
        self.sad_calls = sum(lv["n"] * (N_LIST * N_PASS * len(self.pattern) + N_LIST * len(HALF_PEL)) for lv in self.lv.values())
        # algorithmic bytes of the SAD kernel, SURVEY.md 8(d): 4*w*h + 4 per table call
        self.sad_bytes = {S: lv["n"] * (N_LIST * N_PASS * len(self.pattern) + N_LIST * len(HALF_PEL)) * (4 * S * S + 4)
                          for S, lv in self.lv.items()}
        self.sad_events = []

    def _level_streams(self):
        """one stream per CU level.  XEVE_HIP_LEVEL_PRIO=1: the levels with few, long serial chains (64x64, 32x32) on high-priority streams, so that their short
        kernels are not queued behind the bulk levels' grids"""
        prio = os.environ.get("XEVE_HIP_LEVEL_PRIO", "0")
        if prio == "0":
            return {S: torch.cuda.Stream(device=self.dev) for S in self.sizes}
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        return {S: torch.cuda.Stream(device=self.dev, priority=(hi if S >= (64 if prio == "1" else 32) else lo)) for S in self.sizes}

    def _level(self, S, rng):
        dev, W, H = self.dev, self.W, self.H
        nx, ny = W // S, H // S
        ys, xs = np.meshgrid(np.arange(ny) * S, np.arange(nx) * S, indexing="ij")
        ys, xs = ys.ravel(), xs.ravel()
        n = len(xs)
        off_l = (PAD_L + ys) * self.s_l + PAD_L + xs
        Sc = S // 2
        off_c = (PAD_C + ys // 2) * self.s_c + PAD_C + xs // 2
        lv = dict(n=n, S=S, off_l=torch.from_numpy(off_l.astype(np.int32)).to(dev), off_c=torch.from_numpy(off_c.astype(np.int32)).to(dev))
        # A: search centres per list and pass
        lv["me_jobs"] = []
        for _ in range(N_LIST * N_PASS):
            mvx, mvy = rng.integers(-MV_RANGE, MV_RANGE + 1, n), rng.integers(-MV_RANGE, MV_RANGE + 1, n)
            o1, o2 = off_l, off_l + mvy * self.s_l + mvx
            if SORT_JOBS:
                # launch order = raster order of the SEARCH CENTRES in (8-row, 64-pel = one 128-byte line) bins, so
                # that the waves of a workgroup (consecutive jobs) read the same reference lines and hit in L1
                cy, cx = o2 // self.s_l, o2 % self.s_l
                order = np.lexsort((cx // 64, cy // 8))
                o1, o2 = o1[order], o2[order]
            lv["me_jobs"].append(D.make_jobs(o1, o2, dev))
        lv["sad_out"] = torch.empty((n, len(self.pattern)), dtype=torch.int32, device=dev)
        # B: half-pel interpolation jobs into a dense prediction buffer, then SAD org-vs-dense
        dense = np.arange(n) * S * S
        lv["pred_l"] = [torch.empty((n, S, S), dtype=torch.int16, device=dev) for _ in range(2)]
        lv["pred_c"] = [torch.empty((n, Sc, Sc), dtype=torch.int16, device=dev) for _ in range(4)]
        lv["dense_jobs"] = D.make_jobs(off_l, dense, dev)
        lv["dense_jobs_c"] = D.make_jobs(off_c, np.arange(n) * Sc * Sc, dev)
        lv["sad1"] = torch.empty((n, 1), dtype=torch.int32, device=dev)
        lv["hp_jobs"] = []
        for _ in range(N_LIST):
            mvx, mvy = rng.integers(-MV_RANGE, MV_RANGE + 1, n), rng.integers(-MV_RANGE, MV_RANGE + 1, n)
            per = []
            for hx, hy in HALF_PEL:
                gx, gy = (PAD_L + xs + mvx) * 16 + hx, (PAD_L + ys + mvy) * 16 + hy
                frac = ((gx & 15) != 0).astype(np.int32) | (((gy & 15) != 0).astype(np.int32) << 1)
                per.append(D.make_mc_jobs(gx, gy, dense, frac, dev))
            lv["hp_jobs"].append(per)
        # C/D: quarter-pel motion for merge candidates and the final bi-prediction (any of the 16 phases)
        def qpel_jobs(true_motion=None):
            mvx, mvy = rng.integers(-MV_RANGE * 4, MV_RANGE * 4 + 1, n), rng.integers(-MV_RANGE * 4, MV_RANGE * 4 + 1, n)  # 1/4 pel
            if true_motion is not None:
                mvx, mvy = np.full(n, true_motion[0]), np.full(n, true_motion[1])
            gx, gy = (PAD_L + xs) * 16 + mvx * 4, (PAD_L + ys) * 16 + mvy * 4
            fl = ((gx & 15) != 0).astype(np.int32) | (((gy & 15) != 0).astype(np.int32) << 1)
            # chroma position in 1/32 pel = luma 1/16-pel position relative to the chroma plane origin (xeve_mc.c:487-488)
            cx, cy = (PAD_C + xs // 2) * 32 + mvx * 4, (PAD_C + ys // 2) * 32 + mvy * 4
            fc = ((cx & 31) != 0).astype(np.int32) | (((cy & 31) != 0).astype(np.int32) << 1)
            return (D.make_mc_jobs(gx, gy, dense, fl, dev), D.make_mc_jobs(cx, cy, np.arange(n) * Sc * Sc, fc, dev))
        def with_org_off(j, off):
            j = j.clone()
            j[:, 2] = off
            return j
        lv["hp_jobs_org"] = [[with_org_off(j, lv["off_l"]) for j in per] for per in lv["hp_jobs"]]
        lv["ssd1"] = torch.empty(n, dtype=torch.int64, device=dev)
        lv["merge_jobs"] = [qpel_jobs() for _ in range(N_MERGE)]
        lv["merge_jobs_org"] = [(with_org_off(jl, lv["off_l"]), with_org_off(jc, lv["off_c"])) for jl, jc in lv["merge_jobs"]]
        lv["final_jobs"] = [qpel_jobs(None if self.content == "iid" else ((12, 4), (-12, -4))[l]) for l in range(N_LIST)]
        lv["resi"] = [torch.empty((n, S * S), dtype=torch.int16, device=dev)] + [torch.empty((n, Sc * Sc), dtype=torch.int16, device=dev) for _ in range(2)]
        # quantised levels of Y, U, V in ONE buffer (the bit-count jobs address all three components of a CU by offset)
        flat = torch.empty(n * (S * S + 2 * Sc * Sc), dtype=torch.int16, device=dev)
        lv["coef_flat"] = flat
        lv["coef"] = [flat[:n * S * S].view(n, S * S), flat[n * S * S:n * (S * S + Sc * Sc)].view(n, Sc * Sc), flat[n * (S * S + Sc * Sc):].view(n, Sc * Sc)]
        lv["rec"] = [torch.zeros_like(p) for p in self.org]
        lv["nnz"] = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
        lv["ssd2"] = [torch.empty((n, 2), dtype=torch.int64, device=dev) for _ in range(3)]
        return lv

    # ------------------------------------------------------------------------------------------------------
    def run(self, time_sad=False, only=None):
        bd, qp, s_l, s_c = self.bd, self.qp, self.s_l, self.s_c
        org = self.org
        for S in self.sizes:
            lv = self.lv[S]
            Sc, l2, l2c = S // 2, S.bit_length() - 1, S.bit_length() - 2
            if time_sad:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            # A. integer motion search rounds
            for i, jobs in enumerate(lv["me_jobs"] if only in (None, "A") else ()):
                D.sad_jobs_dual(org[0], s_l, self.ref[i % N_LIST][0], self.ref_s1[i % N_LIST], s_l, jobs, self.cand_l, S, S, bd,
                                out=lv["sad_out"])
            if time_sad:
                e1.record()
                self.sad_events.append((S, "me", e0, e1))
            # B. half-pel refinement: interpolation + SAD fused (the prediction never leaves the CU)
            for l in range(N_LIST if only in (None, "B") else 0):
                for jobs in lv["hp_jobs_org"][l]:
                    D.mc_l_sad_jobs(self.ref[l][0], s_l, org[0], s_l, jobs, S, S, bd, lv["sad1"])
            # C. skip / merge candidates: interpolation + SSD fused, Y / U / V
            for jl, jc in (lv["merge_jobs_org"] if only in (None, "C") else ()):
                D.mc_ssd_jobs(True, self.ref[0][0], s_l, org[0], s_l, jl, S, S, bd, lv["ssd1"])
                for c in (1, 2):
                    D.mc_ssd_jobs(False, self.ref[0][c], s_c, org[c], s_c, jc, Sc, Sc, bd, lv["ssd1"])
            if only not in (None, "D", "D1", "D2", "E"):
                continue
            # D. residual RDO of the (bi-predicted) winner
            for l in range(N_LIST if only in (None, "D", "D1") else 0):
                jl, jc = lv["final_jobs"][l]
                D.mc_jobs(True, self.ref[l][0], s_l, lv["pred_l"][l], S, jl, S, S, bd)
                for c in (1, 2):
                    D.mc_jobs(False, self.ref[l][c], s_c, lv["pred_c"][2 * l + c - 1], Sc, jc, Sc, Sc, bd)
            if only in (None, "D", "D1"):
                D.avg(lv["pred_l"][0].view(-1), lv["pred_l"][1].view(-1), out=lv["pred_l"][0].view(-1))
            for c in ((1, 2) if only in (None, "D", "D1") else ()):
                D.avg(lv["pred_c"][c - 1].view(-1), lv["pred_c"][2 + c - 1].view(-1), out=lv["pred_c"][c - 1].view(-1))
            for c in (range(3) if only in (None, "D", "D2") else ()):
                w, lg, st = (S, l2, s_l) if c == 0 else (Sc, l2c, s_c)
                pred = lv["pred_l"][0] if c == 0 else lv["pred_c"][c - 1]
                dj = lv["dense_jobs"] if c == 0 else lv["dense_jobs_c"]
                if USE_RDOQ:
                    # the chain as preset medium configures it (rdoq = 1, xeve_enc.c:2469): DIFF, SSD, DCT | zero pre-test +
                    # RDOQ (parallel scan) | dequant, IDCT, recon, SSD
                    D.residual_rdoq(org[c], st, pred, w, dj, lg, lg, bd, qp, False, self.lam, c == 0, self.rdoq_est, lv["coef"][c], lv["rec"][c], st,
                                    lv["nnz"][c], lv["ssd2"][c])
                else:
                    # plain quantiser (rdoq = 0): the whole chain in one fused launch
                    D.residual_rdo(org[c], st, pred, w, dj, lg, lg, bd, qp, False, True, lv["coef"][c], lv["rec"][c], st, lv["nnz"][c], lv["ssd2"][c])
            # E. intra gate
            if only in (None, "E"):
                D.satd_jobs(org[0], s_l, lv["pred_l"][0], S, lv["dense_jobs"], self.zero_cand, S, S, bd)

    # ------------------------------------------------------------------------------------------------------
    # F. rate term of pinter_residue_rdo (xeve_pinter.c:1103-1260): CABAC bit counts of the quantised CU left by phase D.
    # Per CU the reference counts: the all-zero alternative, the CU as quantised, and every component with and without its
    # coefficients (xeve_rdo_bit_cnt_cu_inter x2, xeve_rdo_bit_cnt_cu_inter_comp x6) = RATE_JOBS jobs.  (The reference
    # threads the coder state from one component test to the next; here all eight start from the CU's entry state.)
    RATE_JOBS = 8

    def _rate_setup(self, lv):
        from . import lib
        S, n, dev = lv["S"], lv["n"], self.dev
        Sc, ny, nc = S // 2, S * S, S * S // 4
        j = np.zeros((n, self.RATE_JOBS), lib.CU_BITS_JOB_DTYPE)
        cu = np.arange(n)
        j["coef_off"][:, :, 0], j["coef_off"][:, :, 1], j["coef_off"][:, :, 2] = (cu * ny)[:, None], (n * ny + cu * nc)[:, None], (n * (ny + nc) + cu * nc)[:, None]
        j["mode"] = np.array([0, 0, 1, 1, 2, 2, 3, 3], np.uint8)[None, :]
        rng = np.random.default_rng(9)
        j["refi"] = 0  # bi-prediction from reference 0 of both lists
        j["mvd"] = rng.integers(-16, 17, size=(n, 1, 2, 2))
        j["mvp_idx"] = rng.integers(0, 4, size=(n, 1, 2))
        # which components keep their coefficient count in each of the eight jobs
        keep = np.array([[0, 0, 0], [1, 1, 1], [0, 1, 1], [1, 1, 1], [1, 0, 1], [1, 1, 1], [1, 1, 0], [1, 1, 1]], np.int32)
        st = np.zeros(1, lib.SBAC_DTYPE)
        st["range"], st["ctx"] = 16384, 512  # xeve_sbac_reset: the state at the start of a slice
        p = lib.CuBitsParams()
        p.log2_cuw = p.log2_cuh = S.bit_length() - 1
        p.slice_type, p.cm_init, p.chroma_format_idc = 0, 0, 1
        p.num_refp[0] = p.num_refp[1] = 2
        r = dict(params=p, jobs=torch.from_numpy(j.reshape(-1).view(np.uint8).copy()).to(dev), keep=torch.from_numpy(keep).to(dev),
                 state=torch.from_numpy(st.view(np.uint8).copy()).to(dev), bits=torch.empty(n * self.RATE_JOBS, dtype=torch.int32, device=dev))
        need = lib.load().xeve_hip_cu_bits_workspace(n * self.RATE_JOBS, lv["coef_flat"].numel())
        r["ws"] = torch.empty(int(need), dtype=torch.uint8, device=dev)
        return r

    def rate(self):
        """phase F for every level; returns {S: int32 tensor [n, RATE_JOBS] of bit counts}.  One stream per level: the large-CU levels are
        latency-bound (few, long bit strings) and hide under the throughput-bound small-CU ones."""
        out = {}
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = self._level_streams()
        for S in sorted(self.sizes, reverse=True):
            lv = self.lv[S]
            if "rate" not in lv:
                lv["rate"] = self._rate_setup(lv)
            r = lv["rate"]
            st = self._side[S]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                # the coefficient counts come from the quantiser on the device: write them into the job records there
                nnz3 = torch.stack(lv["nnz"], dim=1)  # [n, 3]
                r["jobs"].view(torch.int32).view(lv["n"], self.RATE_JOBS, 11)[:, :, 3:6] = nnz3[:, None, :] * r["keep"][None, :, :]
                D.cu_bits_jobs(lv["coef_flat"], r["state"], r["jobs"], r["params"], want_state=False, workspace=r["ws"], bits=r["bits"])
            out[S] = r["bits"].view(lv["n"], self.RATE_JOBS)
        for S in self.sizes:
            main.wait_stream(self._side[S])
        return out

    # ------------------------------------------------------------------------------------------------------
    # G. pinter_residue_rdo end to end (xeve_hip_residue_rdo_jobs): one bi-predicted candidate per CU of every level, with the
    # vectors phase D uses -- prediction, residual chain with RDOQ from the entry coder state, bit-count rounds, cbf decision.
    def _rdo_setup(self, lv):
        from . import lib
        S, n, dev = lv["S"], lv["n"], self.dev
        j = np.zeros(n, lib.RDO_JOB_DTYPE)
        nx = self.W // S
        idx = np.arange(n)
        j["x"], j["y"] = (idx % nx) * S, (idx // nx) * S
        mv = ((12, 4), (-12, -4)) if self.content != "iid" else None
        rng = np.random.default_rng(17)
        for l in range(N_LIST):
            j["mv"][:, l] = mv[l] if mv else rng.integers(-MV_RANGE * 4, MV_RANGE * 4 + 1, size=(n, 2))
        j["mvd"] = rng.integers(-16, 17, size=(n, 2, 2))
        j["mvp_idx"] = rng.integers(0, 4, size=(n, 2))
        st = np.zeros(1, lib.SBAC_DTYPE)
        st["range"], st["ctx"] = 16384, 512
        p = lib.RdoParams()
        p.log2_cuw = p.log2_cuh = S.bit_length() - 1
        p.pic_w, p.pic_h, p.slice_type, p.chroma_format_idc, p.bit_depth, p.tool_iqt = self.W, self.H, 0, 1, self.bd, 0
        p.num_refp[0] = p.num_refp[1] = 1
        p.qp[0] = p.qp[1] = p.qp[2] = self.qp
        p.lambda_[0] = p.lambda_[1] = p.lambda_[2] = self.lam
        p.dist_chroma_weight[0] = p.dist_chroma_weight[1] = 1.0
        refp = np.zeros(2, lib.REFPIC_DTYPE)  # [refi 0][list 0 / 1]
        for l in range(N_LIST):
            r = self.ref[l]
            refp["y"][l], refp["u"][l], refp["v"][l] = (r[0].data_ptr() + 2 * (PAD_L * self.s_l + PAD_L), r[1].data_ptr() + 2 * (PAD_C * self.s_c + PAD_C),
                                                        r[2].data_ptr() + 2 * (PAD_C * self.s_c + PAD_C))
            refp["poc"][l] = 2 * l
        org = [self.org[0].data_ptr() + 2 * (PAD_L * self.s_l + PAD_L), self.org[1].data_ptr() + 2 * (PAD_C * self.s_c + PAD_C),
               self.org[2].data_ptr() + 2 * (PAD_C * self.s_c + PAD_C)]
        import ctypes as C
        need = lib.load().xeve_hip_residue_rdo_workspace(n, 1, C.byref(p), self.s_l, self.s_c)
        return dict(params=p, jobs=torch.from_numpy(j.view(np.uint8).copy()).to(dev), state=torch.from_numpy(st.view(np.uint8).copy()).to(dev), refp=refp,
                    org=org, ws=torch.empty(int(need), dtype=torch.uint8, device=dev))

    def rdo(self):
        """phase G for every level; returns {S: (results uint8 [n, 72], coef, best)}"""
        # The levels are independent and the large-CU ones are latency-bound (few, long bit strings): each level runs on its own stream so
        # that the 64x64 and 32x32 levels' serial coder chains hide under the throughput-bound small-CU levels.
        out = {}
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = self._level_streams()
        for S in sorted(self.sizes, reverse=True):  # longest chains first
            lv = self.lv[S]
            if "rdo" not in lv:
                lv["rdo"] = self._rdo_setup(lv)
            r = lv["rdo"]
            st = self._side[S]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                out[S] = D.residue_rdo_jobs(r["org"], self.s_l, self.s_c, r["refp"], self.s_l, self.s_c, r["state"], r["params"], r["jobs"], workspace=r["ws"])
        for S in self.sizes:
            main.wait_stream(self._side[S])
        return out

    def _inter_setup(self, lv):
        from . import lib
        import ctypes as C
        S, n, dev = lv["S"], lv["n"], self.dev
        r = lv["rdo"] if "rdo" in lv else self._rdo_setup(lv)
        lv["rdo"] = r
        P = lib.InterParams()
        C.memmove(C.byref(P.rdo), C.byref(r["params"]), C.sizeof(P.rdo))
        lam_mv = int(np.floor(65536.0 * np.sqrt(self.lam)))  # pi->lambda_mv (xeve_pinter.c:1763)
        P.me.me.lambda_mv, P.me.me.faststep, P.me.me.max_search_range = lam_mv, 3, 64
        P.me.me.min_clip[0], P.me.me.min_clip[1], P.me.me.max_clip[0], P.me.me.max_clip[1] = -127, -127, self.W - 1, self.H - 1  # xeve_pinter.c:2124-2127
        P.me.hpel_cnt, P.me.qpel_cnt = 8, 8
        for l in range(2):
            P.refi_bits[l][0], P.range_recentre[l][0] = 0, 16  # one reference picture per list, one picture away (gop 8: 64 / 8 -> clipped to 64 >> 2)
        P.max_cand, P.poc, P.col_list_poc0, P.skip_th = 3, 1, 0, 0.0
        j = np.zeros(n, lib.INTER_JOB_DTYPE)
        nx = self.W // S
        idx = np.arange(n)
        j["x"], j["y"] = (idx % nx) * S, (idx // nx) * S
        rng = np.random.default_rng(23)
        # merge / MVP candidates as a coded neighbourhood gives them: mostly the true motion (+-12, +-4 quarter pel for the structured picture),
        # sometimes a slightly different vector, sometimes the (1, 1) of an unavailable neighbour; random on the i.i.d. picture
        for l in range(N_LIST):
            true = np.array(((12, 4), (-12, -4))[l]) if self.content != "iid" else None
            for k in range(4):
                kind = rng.integers(0, 10, size=n)
                v = np.tile(true, (n, 1)) if true is not None else rng.integers(-MV_RANGE * 4, MV_RANGE * 4 + 1, size=(n, 2))
                v = np.where((kind == 7)[:, None], v + rng.integers(-6, 7, size=(n, 2)), v)
                v = np.where((kind >= 8)[:, None], 1, v)
                j["mvp"][:, l, k] = v
        j["mv_col"] = rng.integers(-8, 9, size=(n, 2))
        refp = r["refp"].copy()
        refp["poc"][0], refp["poc"][1] = 0, 2
        need = lib.load().xeve_hip_pinter_analyze_cu_workspace(n, 1, C.byref(P), self.s_l, self.s_c)
        return dict(params=P, jobs=torch.from_numpy(j.view(np.uint8).copy()).to(dev), refp=refp, ws=torch.empty(int(need), dtype=torch.uint8, device=dev))

    def inter(self):
        """phase H: the whole inter analysis (xeve_pinter_analyze_cu) of every CU of every level, one stream per level; returns {S: results uint8 [n, 96]}"""
        out = {}
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = self._level_streams()
        for S in self.sizes:
            if "inter" not in self.lv[S]:
                self.lv[S]["inter"] = self._inter_setup(self.lv[S])

        def level(S):
            lv = self.lv[S]
            h, r = lv["inter"], lv["rdo"]
            with torch.cuda.stream(self._side[S]):
                return D.pinter_analyze_cu_jobs(r["org"], self.s_l, self.s_c, h["refp"], self.s_l, self.s_c, r["state"], h["params"], h["jobs"], workspace=h["ws"])[0]
        order = sorted(self.sizes, reverse=True)
        for S in order:
            self._side[S].wait_stream(main)
        # XEVE_HIP_LEVEL_THREADS=1: one host thread per level (the library call releases the GIL), the four launch sequences issued side by side.  MEASURED
        # (gpurun_out/r02c12, 4K i.i.d.): 46.4 ms against 45.6 ms from one thread -- the step is bound by the GPU (the bit counter), not by the host's issue
        # order, although a kernel trace taken UNDER rocprofv3 suggests otherwise (the profiler's per-launch cost delays the later levels' first kernels by
        # 17 ms); GPU_MAX_HW_QUEUES=8 (a hardware queue per level instead of three for four) makes it worse, 52.2 ms.  Off by default.
        if self.level_threads and len(order) > 1 and not torch.cuda.is_current_stream_capturing():
            if not hasattr(self, "_pool"):
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=len(order))
            dev_idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()

            def worker(S):
                torch.cuda.set_device(dev_idx)
                return level(S)
            for S, f in [(S, self._pool.submit(worker, S)) for S in order]:
                out[S] = f.result()
        else:
            for S in order:
                out[S] = level(S)
        for S in self.sizes:
            main.wait_stream(self._side[S])
        return out

    # ------------------------------------------------------------------------------------------------------
    # I. the intra analysis (xeve_hip_pintra_analyze_cu_jobs) of every CU of the levels 64 .. 8 and of every 4x4 CU: neighbours from a reconstruction of
    # the picture (reference picture 0 of list 0 stands in for PIC_MODE), every 4x4 unit coded and intra with a random luma mode, no inter candidate to
    # prune against (I picture: all five predictors go through the luma RDO -- the most work the analysis can be asked for).
    INTRA_SIZES = (64, 32, 16, 8, 4)

    def _intra_setup(self, S):
        from . import lib
        import ctypes as C
        dev = self.dev
        nx, ny = self.W // S, self.H // S
        n = nx * ny
        j = np.zeros(n, lib.INTRA_JOB_DTYPE)
        idx = np.arange(n)
        j["x"], j["y"] = (idx % nx) * S, (idx // nx) * S
        j["inter_satd"] = 0xFFFFFFFF
        P = lib.IntraParams()
        P.log2_cuw = P.log2_cuh = S.bit_length() - 1
        P.w_scu, P.h_scu, P.slice_type, P.chroma_format_idc, P.bit_depth, P.tool_iqt, P.constrained_intra_pred = self.W // 4, self.H // 4, 2, 1, self.bd, 0, 0
        P.qp[0] = P.qp[1] = P.qp[2] = self.qp
        P.lambda_[0] = P.lambda_[1] = P.lambda_[2] = self.lam
        P.sqrt_lambda0 = float(np.sqrt(self.lam))
        P.dist_chroma_weight[0] = P.dist_chroma_weight[1] = 1.0
        if not hasattr(self, "_intra_maps"):
            nu = (self.W // 4) * (self.H // 4)
            rng = np.random.default_rng(29)
            st = np.zeros(1, lib.SBAC_DTYPE)
            st["range"], st["ctx"] = 16384, 512
            self._intra_maps = dict(scu=torch.full((nu,), -(1 << 31) | (1 << 15), dtype=torch.int32, device=dev),  # MCU COD | IF
                                    ipm=torch.from_numpy(rng.integers(0, 5, size=nu).astype(np.int8)).to(dev), tidx=torch.zeros(nu, dtype=torch.uint8, device=dev),
                                    state=torch.from_numpy(st.view(np.uint8).copy()).to(dev))
        at = lambda t, pad, s: t.data_ptr() + 2 * (pad * s + pad)
        org = [at(self.org[0], PAD_L, self.s_l), at(self.org[1], PAD_C, self.s_c), at(self.org[2], PAD_C, self.s_c)]
        mod = [at(self.ref[0][0], PAD_L, self.s_l), at(self.ref[0][1], PAD_C, self.s_c), at(self.ref[0][2], PAD_C, self.s_c)]
        need = lib.load().xeve_hip_pintra_analyze_cu_workspace(n, 1, C.byref(P))
        return dict(params=P, jobs=torch.from_numpy(j.view(np.uint8).copy()).to(dev), org=org, mod=mod, n=n, ws=torch.empty(int(need), dtype=torch.uint8, device=dev))

    def intra(self, sizes=None):
        """phase I: the intra analysis of every CU of every level, one stream per level; returns {S: (results uint8 [n, 32], coef, rec, best)}"""
        out = {}
        sizes = tuple(sizes or self.INTRA_SIZES)
        main = torch.cuda.current_stream()
        if not hasattr(self, "_iside"):
            self._iside, self._ilv = {S: torch.cuda.Stream(device=self.dev) for S in self.INTRA_SIZES}, {}
        for S in sizes:
            if S not in self._ilv:
                self._ilv[S] = self._intra_setup(S)
            h, m = self._ilv[S], self._intra_maps
            st = self._iside[S]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                out[S] = D.pintra_analyze_cu_jobs(h["org"], self.s_l, self.s_c, h["mod"], self.s_l, self.s_c, m["scu"], m["ipm"], m["tidx"], m["state"], h["params"], h["jobs"],
                                                  workspace=h["ws"])
        for S in sizes:
            main.wait_stream(self._iside[S])
        return out

    def capture(self):
        """Record one pass into a HIP graph (all launches of run() go to torch's current stream, which is the capture
        stream here); replay() then re-issues the ~150 launches with one host call.  Matters for small pictures, where
        the eager pass is bound by host launch overhead rather than by the GPU."""
        self.run()  # warm-up outside capture: allocator pools, lazy module loading
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.run()
        return self.graph

    def replay(self):
        self.graph.replay()

    def sad_time_ms(self):
        """sum of HIP-event durations of the integer-search SAD launches recorded by run(time_sad=True), per size"""
        torch.cuda.synchronize()
        out = {}
        for S, _, e0, e1 in self.sad_events:
            out[S] = out.get(S, 0.0) + e0.elapsed_time(e1)
        return out


def D_same_jobs(lv, c):
    key = "same_jobs_%d" % c
    if key not in lv:
        off = lv["off_l"] if c == 0 else lv["off_c"]
        lv[key] = torch.stack([off, off], dim=1).contiguous()
    return lv[key]


class CtuWalkIntra:
    """The CTU mode decision of I pictures on the device (xeve_hip_mode_analyze_ctu_jobs), chains = pictures in lockstep: every chain is a 128x128 picture of its
    own (four CTUs); a step decides the same CTU of every picture -- the quad-tree 64 .. 4 with the intra analysis of every node, the maps and the reconstruction
    updated CU by CU.  content "noise": every node is visited and decided; "smooth": the early termination of I pictures prunes the tree."""

    CTUS = [(0, 0), (64, 0), (0, 64), (64, 64)]

    def __init__(self, chains, device, content="noise", seed=7, qp8=32, bit_depth=10, max_cu=32, write=False):
        import ctypes as C

        from . import lib
        self.n, self.dev, w = chains, device, 128
        P = lib.TreeParams()
        qp = qp8 + 6 * (bit_depth - 8)
        P.ip.w_scu, P.ip.h_scu, P.ip.slice_type, P.ip.chroma_format_idc, P.ip.bit_depth = w // 4, w // 4, 2, 1, bit_depth
        P.ip.qp[0], P.ip.qp[1], P.ip.qp[2] = qp, qp - 1, qp - 2
        lam = 0.57 * 2.0 ** ((qp8 - 12) / 3.0)
        P.ip.lambda_[0], P.ip.sqrt_lambda0 = lam, lam ** 0.5
        P.ip.dist_chroma_weight[0], P.ip.dist_chroma_weight[1] = 2.0 ** (1 / 3.0), 2.0 ** (2 / 3.0)
        P.ip.lambda_[1], P.ip.lambda_[2] = lam / P.ip.dist_chroma_weight[0], lam / P.ip.dist_chroma_weight[1]
        P.pic_w, P.pic_h, P.log2_ctu, P.max_cu, P.min_cu, P.min_cuwh, P.slice_qp = w, w, 6, max_cu, 4, 4, qp
        self.P, self.w = P, w
        g = torch.Generator(device=device).manual_seed(seed)
        n, maxv = chains, (1 << bit_depth) - 1
        if content == "noise":
            self.org = [torch.randint(0, maxv + 1, (n, w >> s, w >> s), device=device, generator=g, dtype=torch.int16) for s in (0, 1, 1)]
        else:  # "smooth": the early termination prunes at 32x32; "texture": detail the predictors cannot follow, small levels -- the tree goes down to 4x4 like on noise
            yy, xx = torch.meshgrid(torch.arange(w, device=device), torch.arange(w, device=device), indexing="ij")
            base = (maxv / 2) * (1 + 0.6 * torch.sin(xx / 19.0) * torch.cos(yy / 23.0))
            if content == "texture":
                base = base + (maxv / 16) * torch.sin(xx / 1.3 + yy / 2.9) * torch.cos(xx / 3.1 - yy / 1.7)
            amp = 3 if content == "smooth" else 12
            luma = (base[None] + torch.randint(-amp, amp + 1, (n, w, w), device=device, generator=g)).clamp(0, maxv).to(torch.int16)
            self.org = [luma, luma[:, ::2, ::2].contiguous(), luma[:, ::2, ::2].contiguous()]
        nscu = (w // 4) ** 2
        self.mod = [torch.full_like(t, 1 << (bit_depth - 1)) for t in self.org]
        self.ms, self.mc = torch.zeros((n, nscu), dtype=torch.int32, device=device), torch.zeros((n, nscu), dtype=torch.int32, device=device)
        self.mi, self.mt = torch.zeros((n, nscu), dtype=torch.int8, device=device), torch.zeros((n, nscu), dtype=torch.uint8, device=device)
        st = np.zeros(n, lib.SBAC_DTYPE)
        st["range"], st["code_bits"], st["ctx"] = 16384, 11, 512  # the coder state at the start of a slice
        self.states = torch.from_numpy(st.view(np.uint8).copy()).to(device)
        self.need = int(lib.load().xeve_hip_mode_analyze_ctu_workspace(n, C.byref(P), None, w, w // 2))
        self.ws = torch.empty(self.need, dtype=torch.uint8, device=device)
        self.pe = (self.org[0][0].numel(), self.org[1][0].numel(), self.mod[0][0].numel(), self.mod[1][0].numel(), nscu)
        self.jobs = []
        for (x, y) in self.CTUS:
            j = np.zeros(n, lib.CTU_JOB_DTYPE)
            j["x"], j["y"], j["sbac"], j["pic"] = x, y, np.arange(n), np.arange(n)
            self.jobs.append(torch.from_numpy(j.view(np.uint8).copy()).to(device))
        self.job = self.jobs[0].clone()  # the call's operands stay at fixed addresses: the library replays the walk from a HIP graph
        self.outputs = (torch.zeros((n, lib.CTU_DATA_BYTES), dtype=torch.uint8, device=device), torch.zeros((n, D.SBAC_BYTES), dtype=torch.uint8, device=device),
                        torch.zeros(n, dtype=torch.float64, device=device))
        self.stream = torch.cuda.Stream(device=device)  # (a graph cannot be captured on the default stream)
        # write=True: every decided CTU is also WRITTEN on the device (xeve_hip_eco_ctu_jobs) -- the coder state of a chain is then the writer's, advanced in place, and
        # what a step produces is slice data
        self.write = write
        if write:
            EP = lib.EcoParams()
            EP.chroma_format_idc, EP.slice_type, EP.log2_ctu, EP.pic_w, EP.pic_h, EP.w_scu, EP.h_scu = 1, 2, 6, w, w, w // 4, w // 4
            self.EP = EP
            self.bytes = (torch.zeros((n, 1 << 14), dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.int32, device=device))
            self.nscu = nscu
        self.k, self.out = 0, None

    def step(self):
        """decides CTU (k mod 4) of every picture; the coder state of each chain carries over"""
        w = self.w
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.job.copy_(self.jobs[self.k % 4])
            self.out, nxt, self.cost = D.mode_analyze_ctu_jobs([t.data_ptr() for t in self.org], w, w // 2, [t.data_ptr() for t in self.mod], w, w // 2, self.ms, self.mi, self.mt,
                                                               self.mc, self.states, self.P, self.job, pic_elems=self.pe, workspace=self.ws, outputs=self.outputs)
            if self.write:
                D.eco_ctu_jobs(self.out, self.states, self.EP, self.ms, self.mi, self.mt, self.mc, self.job, map_pic_elems=self.nscu, out=self.bytes)
            else:
                self.states.view(-1).copy_(nxt.view(-1))
        torch.cuda.current_stream().wait_stream(self.stream)
        self.k += 1

    def mean_depth(self):
        from . import lib
        return float(self.out.cpu().numpy().reshape(-1).view(np.dtype(lib.CTU_DATA_DTYPE))["depth"].mean())
