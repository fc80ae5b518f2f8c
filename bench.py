#!/usr/bin/env python3
"""bench.py -- XEVE's inter analysis (the inter-prediction / RDO hot path) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the implemented hot path over ONE 3840x2160 Baseline-medium B picture of synthetic i.i.d. 8-bit
content (<< 2, the encoder's 10-bit internal depth), every plane already resident in HBM: the whole of
`xeve_pinter_analyze_cu` (skip / merge analysis, temporal direct, both lists' motion searches, check_best_mvp, the iterated
bi-prediction search, every pinter_residue_rdo with RDOQ and real CABAC bit counting, the mode decision, reconstruction and
exit coder state) for EVERY CU of EVERY quad-tree level 64 .. 8 (`HotPathPass.inter()`, xeve_amd/workload.py; one stream per
level).  `value` is pictures per second over all ranks.  It is NOT an encode rate: the quad-tree decision, intra analysis and
the bitstream writer that turn these per-CU results into a bitstream run in the reference encoder, and
the encoder-in-the-loop rate is reported separately by the e2e tests.  Each rank owns its own closed GOP (its own pictures),
per-GPU work is fixed ("weak"), and the data path has no collective.

The JSON line also carries
  roofline     : the SAD kernel on that path (k_me_epzs: the integer motion search) -- algorithmic bytes (256 per 64 sample
                 pairs) over its HIP-event time measured live in the timed region on the launch streams, the
                 ceiling the counters say binds it, and the physical HBM / L2 / LDS figures of the PMC passes in profiles/;
  kernels      : per kernel class, HIP-event time per picture (search, sub-pel, CABAC bit counting, prediction, residual
                 chain, RDOQ) from an untimed pass with every class's timer on;
  secondary    : the same step on SURVEY 8(d)'s structured input, at 1920x1080, the round-1 synthetic-vector pass A..E, the intra analysis of every CU, the CTU mode
                 decision of I pictures on the device (chains in lockstep);
  cpu_baseline : the reference encoder itself (oracle/_ref/xeveb_app, compiled in place from the reference) on this box's
                 host cores, -m 8 and -m 1, on the first frames of the same kind of input -- rank 0, N = 1 only.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
LDS_PEAK_GBS = 150000.0  # ds_read_b64 / b128, every CU streaming
VALU_SAD_PEAK_GBS = 256 * 4 * 32 * 8 * 2.4  # v_sad_u16: 2 sample pairs = 8 algorithmic bytes per lane, 32 lanes / clk / SIMD
BYTES_PER_SEARCH_UNIT = 256  # 64 sample pairs x (2 + 2) bytes (4*w*h per block SAD; the +4 result bytes are dropped)


def pmc_summary(name="r02_search_pmc.json"):
    """per-launch PMC figures kept under profiles/ (separate --pmc passes, as the PMC rules require; tools/gpu/r02_pmc.sh + tools/make_pmc_profiles.py); None when absent"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "usable_cores": len(os.sched_getaffinity(0))}


class CpuApp:
    """the reference encoder on this box's host cores: xeveb_app -m 8 and -m 1 side by side on the first `frames` frames of a seeded random 8-bit 4:2:0
    clip (started in the background while the GPU runs the secondary measurements; `result()` waits for them)"""

    def __init__(self, width, height, frames=2):
        self.w, self.h, self.frames, self.procs, self.err = width, height, frames, {}, None
        self.exe = os.path.join(ROOT, "oracle", "_ref", "xeveb_app")
        if not os.path.exists(self.exe):
            self.err = "oracle/_ref/xeveb_app not built"
            return
        try:
            self.dir = tempfile.mkdtemp(prefix="xeve_bench_")
            yuv = os.path.join(self.dir, "in.yuv")
            np.random.default_rng(4).integers(0, 256, size=width * height * 3 // 2 * frames, dtype=np.uint8).tofile(yuv)
            for m in (8, 1):
                cmd = [self.exe, "-i", yuv, "-w", str(width), "-h", str(height), "-z", "30", "--preset", "medium", "--closed-gop", "-I", "8", "--frames", str(frames),
                       "-m", str(m), "-o", os.path.join(self.dir, "m%d.evc" % m)]
                self.procs[m] = (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), time.perf_counter(), cmd)
        except Exception as e:  # noqa: BLE001 -- a reported extra, never a reason to lose the GPU number
            self.err = repr(e)[:200]

    def result(self, timeout=900):
        if self.err:
            return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "failed: " + self.err}
        out = {}
        for m, (p, t0, cmd) in self.procs.items():
            try:
                txt, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                out[m] = {"error": "timeout"}
                continue
            fps = re.search(r"Average encoding speed\s*=\s*([0-9.]+)", txt)  # the app's own figure: times xeve_encode only (app/xeve_app.c:1401)
            cum = [(int(a), float(b)) for a, b in re.findall(r"\[\s*(\d+)\s*/\s*\d+ frames \] \[\s*([0-9.]+) frame/sec", txt)]
            t = [(k + 1) / f for k, f in cum if f > 0]
            out[m] = {"fps": float(fps.group(1)) if fps else None, "wall_s": round(time.perf_counter() - t0, 1),
                      "s_per_frame_in_coding_order": [round(b - a, 2) for a, b in zip([0.0] + t[:-1], t)], "rc": p.returncode}
        try:
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)
        except Exception:
            pass
        h = host_info()
        m8, m1 = out.get(8, {}), out.get(1, {})
        return {"value": m8.get("fps"), "unit": "frames/s", "cores": 8, "kind": "reference",
                "sample": "oracle/_ref/xeveb_app (the unmodified reference, AVX2 dispatch) -w %d -h %d --preset medium --closed-gop -I 8 --frames %d on numpy default_rng(4) "
                          "uniform 8-bit 4:2:0 (frame 0 is the IDR picture, the rest inter pictures); `value` = -m 8 (the library's thread maximum), `m1` = -m 1 (the thread count "
                          "the byte-identical tests use); the two ran side by side on different cores" % (self.w, self.h, self.frames),
                "m8": m8, "m1": m1, "host": h}


def time_steps(fn, steps, sync):
    """wall time of `steps` calls of fn between device fences, in ms per call"""
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--live-prof", default="search", help="kernel classes timed with HIP events INSIDE the timed region (comma list of search, cu_bits; 'none'): "
                    "every timed launch costs two event records on its stream -- MEASURED: with search + cu_bits (140 launches per step) the step is 17 %% slower "
                    "than with none, with the search alone (40 launches, the roofline's kernel) the difference is within the run-to-run noise")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed secondary measurements (structured input, 1080p, synthetic A..E pass)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # harness self-test only: XEVE_BENCH_SHARE_GPU=1 runs every rank on GPU 0 with the gloo backend, so that the N > 1
    # control path (barriers, max-over-ranks, rank-0 JSON) can be exercised on a one-GPU box; never used for numbers
    share = os.environ.get("XEVE_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if not os.path.exists(os.path.join(ROOT, "xeve_amd", "lib", "libxeve_hip.so")) and local == 0:
        import __graft_entry__

        __graft_entry__.build()  # fresh checkout on the GPU box: compile once (hipcc is in the image)
    if world > 1:
        dist.barrier()
    import xeve_amd
    from xeve_amd import lib
    from xeve_amd.workload import N_LIST, N_PASS, HotPathPass

    xeve_amd.init(local)
    solo = rank == 0 and world == 1
    cpu = CpuApp(a.width, a.height) if solo and not a.no_cpu_baseline else None  # (host cores only; runs while the GPU is timed)

    # every rank = one encoder process bound to one GPU working on its own closed GOP (xeve_amd/gop.py)
    wl = HotPathPass(a.width, a.height, dev, seed=4 + rank, content="iid")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.inter()
    fence()
    live_classes = [c for c in a.live_prof.split(",") if c in ("search", "cu_bits")]
    if "cu_bits" in live_classes:
        live_classes.append("cu_bits_slow")
    lib.prof_enable(live_classes or None)  # the roofline's kernel: timed live, on its launch streams
    lib.prof_read()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = wl.inter()
    fence()
    dt = time.perf_counter() - t0
    live = lib.prof_read()
    lib.prof_enable(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def winners(res):
        modes = np.concatenate([v.cpu().numpy().reshape(-1).view(np.dtype(lib.INTER_RESULT_DTYPE))["best_idx"] for v in res.values()])
        c = np.bincount(modes, minlength=5)
        return {"cus": int(len(modes)), "l0": int(c[0]), "l1": int(c[1]), "bi": int(c[2]), "skip": int(c[3]), "direct": int(c[4])}

    line = None
    if rank == 0:
        s_ms, s_n, s_u = live["search"]
        b_ms, b_n, b_u = live["cu_bits"]
        alg = s_u * BYTES_PER_SEARCH_UNIT
        alg_gbs = alg / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
        pmc = pmc_summary()
        avg_launch_s = s_ms * 1e-3 / max(1, s_n)
        roof = {"kernel": "k_me_epzs<8|16|32|64, uni|bi> (the integer motion search of xeve_hip_pinter_analyze_cu_jobs; the SAD kernel of the path)",
                "launches_in_region": s_n, "avg_launch_ms": round(1e3 * avg_launch_s, 4), "algorithmic_bytes_per_launch": int(alg / max(1, s_n)),
                "algorithmic_GBps": round(alg_gbs, 1), "sad_evaluations_as_8x8_tiles_per_picture": int(s_u / a.steps)}
        # `achieved` = algorithmic bytes over the live launch time, against the HBM roof (the metric's "SAD-kernel HBM GB/s vs peak").  Should the algorithmic rate
        # ever pass the HBM peak it is being served on chip, and the fraction is then quoted against the L2 roof instead of pretending to be an HBM figure.
        if alg_gbs <= HBM_PEAK_GBS:
            roof.update({"bound": "hbm", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / HBM_PEAK_GBS, 4)})
        else:
            roof.update({"bound": "l2", "achieved": round(alg_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / L2_PEAK_GBS, 4),
                         "note": "algorithmic rate exceeds the HBM peak (on-chip reuse): quoted against the L2 roof"})
        roof["valu_sad_frac"] = round(alg_gbs / VALU_SAD_PEAK_GBS, 4)
        roof["traffic"] = None
        if pmc:  # physical figures of the same kernel, per launch, from the committed PMC passes (profiles/r02_search_pmc.json)
            roof["traffic"] = pmc.get("hbm_bytes_per_launch_x2")
            t_pmc = pmc.get("avg_launch_s")
            if pmc.get("hbm_bytes_per_launch_x2") and t_pmc:
                roof["hbm_physical_GBps"] = round(pmc["hbm_bytes_per_launch_x2"] / t_pmc / 1e9, 1)
                roof["hbm_physical_frac"] = round(roof["hbm_physical_GBps"] / HBM_PEAK_GBS, 4)
            if pmc.get("l2_read_bytes_per_launch") and t_pmc:
                roof["l2_GBps"] = round(pmc["l2_read_bytes_per_launch"] / t_pmc / 1e9, 1)
                roof["l2_frac"] = round(roof["l2_GBps"] / L2_PEAK_GBS, 4)
            roof["binding"] = ("what the counters say binds the kernel: instruction issue and dependent latency (per 8x8 job ~1 400 VALU + ~700 SALU instructions, 60 % of the wave "
                               "cycles issuing or stalled on issue), not a memory level -- HBM traffic is ~2 % of the algorithmic bytes (every plane is read about once per launch), "
                               "L2 and HBM each run below 5 % of their peaks; see profiles/r02_search_pmc.json")
        mf = pmc_summary("r02_mfma_pmc.json")
        if mf:  # the only MFMA kernels of the path: the fused residual chain of 32x32 / 64x64 blocks (north_star: MFMA utilisation from rocprof against the peak)
            roof["mfma"] = {k: {"avg_launch_us": v["avg_launch_us"], "achieved_TOPS_i8": v["achieved_TOPS"], "peak_TOPS_i8_dense": v["peak_TOPS_i8_dense"],
                                "utilisation": v["mfma_utilisation"], "mfma_busy_over_all_simd_cycles": v["mfma_busy_over_all_simd_cycles"]} for k, v in mf["per_kernel"].items()}
        line = {
            "metric": "encoded frames/sec @ 2160p Baseline medium; SAD-kernel HBM GB/s vs peak",
            "value": round(world * a.steps / dt, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "s16 samples, s32/s64 accumulation, f64 cost comparisons (bit-exact)",
            "data": "synthetic",
            "value_is": "pictures/s of the whole inter analysis (xeve_pinter_analyze_cu of every CU of every level) -- the implemented unit of work; NOT an encode rate",
            "config": {
                "workload": "whole inter analysis (skip / merge, direct, motion search both lists, bi-prediction search, pinter_residue_rdo incl. RDOQ + CABAC bit counts, "
                            "decision, reconstruction) of every CU of the levels 64, 32, 16, 8 of one %dx%d Baseline-medium B picture per step per GPU; i.i.d. uniform 8-bit "
                            "source << 2, one reference picture per list, search range +-64, 3 merge candidates, QP 32" % (a.width, a.height),
                "bit_depth": 10, "qp": 32, "ctu": 64, "cu_sizes": list(wl.sizes), "ref_lists": N_LIST, "cus_per_picture": int(sum(wl.lv[S]["n"] for S in wl.sizes)),
                "winners_last_step": winners(out),
                "parallelism": "closed-GOP shard per GPU, no collectives" + (" [SELF-TEST: all ranks share GPU 0, gloo]" if share else ""),
            },
            "roofline": roof,
            "kernels_in_timed_region": dict(
                [("search", {"ms_per_picture": round(s_ms / a.steps, 3), "launches_per_picture": s_n // a.steps})] +
                ([("cu_bits", {"ms_per_picture": round(b_ms / a.steps, 3), "launches_per_picture": b_n // a.steps, "bins_per_picture": int(b_u / a.steps),
                               "Gbin_per_s": round(b_u / (b_ms * 1e-3) / 1e9, 3) if b_ms > 0 else None,
                               "jobs_on_the_slow_path_per_picture": int(live["cu_bits_slow"][2] / a.steps)})] if "cu_bits" in live_classes else []) +
                [("note", "sums of per-launch HIP-event times on the launch streams (classes named by --live-prof; the others are timed in the untimed pass below: "
                          "`kernels`); the four levels run on four streams, so the sums can exceed the wall time")]),
        }

    if solo and not a.no_secondary:
        sync = torch.cuda.synchronize
        sec = {}
        # (1) every kernel class, untimed pass with all timers on
        lib.prof_enable(lib.PROF_CLASSES)
        lib.prof_read()
        reps = 3
        for _ in range(reps):
            wl.inter()
        allc = lib.prof_read()
        lib.prof_enable(None)
        line["kernels"] = {k: {"ms_per_picture": round(v[0] / reps, 3), "launches_per_picture": v[1] // reps} for k, v in allc.items()}
        cb = allc["cu_bits"]
        line["kernels"]["cu_bits"].update({"bins_per_picture": int(cb[2] / reps), "Gbin_per_s": round(cb[2] / (cb[0] * 1e-3) / 1e9, 3) if cb[0] > 0 else None,
                                           "jobs_on_the_slow_path_per_picture": int(allc["cu_bits_slow"][2] / reps)})
        # (2) the same step on the structured input and at 1920x1080
        del wl.lv
        del wl
        torch.cuda.empty_cache()

        def one(w, h, content, steps):
            p = HotPathPass(w, h, dev, seed=5, content=content)
            p.inter()
            ms = time_steps(p.inter, steps, sync)
            res = {"ms_per_picture": round(ms, 3), "pictures_per_s": round(1e3 / ms, 2), "winners": winners(p.inter())}
            return p, res
        ws, sec["structured_%dx%d" % (a.width, a.height)] = one(a.width, a.height, "structured", 10)
        # (3) the round-1 synthetic-vector pass (phases A..E: fixed 90-candidate rounds, no decisions), kept as a secondary figure
        ws.run()
        ms = time_steps(ws.run, 10, sync)
        sec["synthetic_vector_pass_A_to_E"] = {"ms_per_picture": round(ms, 3), "pictures_per_s": round(1e3 / ms, 2),
                                               "note": "xeve_amd/workload.py run(): fixed candidate pattern x %d lists x %d rounds, half-pel MC+SAD, merge MC+SSD, bi-pred MC, residual "
                                                       "chain with RDOQ, SATD -- kernels of the table layer, not of the inter analysis" % (N_LIST, N_PASS)}
        # (4) the intra analysis (xeve_hip_pintra_analyze_cu_jobs) of every CU of every level 64 .. 4 of the same picture: neighbours from a reconstruction, all
        # five predictors through the luma RDO (I picture: nothing to prune against)
        ws.intra()
        ms = time_steps(ws.intra, 10, sync)
        sec["intra_analysis_structured_%dx%d" % (a.width, a.height)] = {"ms_per_picture": round(ms, 3), "pictures_per_s": round(1e3 / ms, 2),
                                                                        "cus": int(sum(v["n"] for v in ws._ilv.values())),
                                                                        "note": "xeve_amd/workload.py intra(): pintra_analyze_cu of every CU of the levels 64, 32, 16, 8, 4"}
        del ws
        torch.cuda.empty_cache()
        # (5) the caller above the CU on the device: the CTU mode decision of I pictures (quad-tree 64 .. 4, intra analysis of every node, maps + reconstruction updated
        # CU by CU), chains = pictures in lockstep; one chain alone = the latency of one CTU
        from xeve_amd.workload import CtuWalkIntra
        walk = {}
        for chains in (1, 1024):
            wk = CtuWalkIntra(chains, dev, "noise")
            wk.step()
            ms = time_steps(wk.step, 3, sync)
            walk["chains_%d" % chains] = {"ms_per_ctu_step": round(ms, 2), "ctus_per_s": round(chains / ms * 1e3, 1),
                                          "equivalent_%dx%d_pictures_per_s" % (a.width, a.height): round(chains / ms * 1e3 / (((a.width + 63) // 64) * ((a.height + 63) // 64)), 3)}
            del wk
            torch.cuda.empty_cache()
        walk["note"] = ("xeve_amd/workload.py CtuWalkIntra: xeve_hip_mode_analyze_ctu_jobs on i.i.d. content (every node of the tree decided), max_cu_intra 32, min 4; the "
                        "decision of a CTU is serial (neighbours' reconstruction, coder state), so the rate scales with the pictures in flight, not within one")
        sec["ctu_mode_decision_I_pictures"] = walk
        for content in ("iid", "structured"):
            p, sec["%s_1920x1080" % content] = one(1920, 1080, content, 20)
            del p
            torch.cuda.empty_cache()
        line["secondary"] = sec
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu.result()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
