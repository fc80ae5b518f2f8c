#!/usr/bin/env python3
"""bench.py -- encoded frames/s of the closed-GOP batch encoder on MI355X (3840x2160 Baseline preset medium), next to the reference encoder on the host.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

THE JOB (per GPU) is ONE real encode: independent closed GOPs of F frames each (default F = 2: the IDR picture and one inter picture, the sample the CPU baseline
codes too), in --batches batches side by side (the first of G GOPs -- a batch ends at 2^32 original samples --, the others as large as HBM still allows), of synthetic i.i.d. uniform 8-bit 4:2:0 frames, resident in HBM before the clock starts, coded by xeve_hip_enc_* (include/xeve_hip.h) exactly as
`xeveb_app --preset medium --closed-gop -I 8 -m 8` codes them: CTU mode decision (quad-tree, intra + inter analysis with motion search, RDOQ, CABAC bit counts),
entropy writer, loop filter, second writer pass, padding, parameter sets + SEI + slice NAL units.  The first and the last GOP of every batch of rank 0 are the reference's
own seed-4 input: their bitstreams are checked against each other and against the md5 recorded from the unmodified reference (tests/golden/e2e_v1.json) in the same run.

A STEP.  The encode is a sequence of lockstep CTU steps (one CTU of every row chain of every GOP decided and written per step; a picture's set-up rides on its first
step, its end -- loop filter, slice data, NAL units -- on its last).  The job's steps are cut into W + K equal slices: the first W slices are the untimed warm-up, the
K others are timed between device fences + barriers.  `value` = frames coded inside the timed slices (GOPs x F x the timed share of the job) / the timed seconds, over all
ranks; the whole job always runs, so the default and the driver's K / W time the same work.  Every rank encodes its own GOPs ("weak"); no collective in the data path.

The JSON line also carries
  roofline     : the SAD kernel of the path (k_me_epzs, the integer motion search): v_sad_u16 work against the VALU roof (what binds a search that re-reads its window
                 from cache), with the algorithmic-bytes-over-HBM-peak figure of BASELINE's metric and the physical HBM traffic of the PMC passes as secondary keys;
                 `by_time` = the kernel class that dominates the GPU time (CABAC bit counting) against an instruction-issue roof;
  cpu_baseline : oracle/_ref/xeveb_app (the unmodified reference, compiled in place) on this box's host cores, -m 8 and -m 1, same picture size, same kind of input.
"""
import argparse
import hashlib
import json
import os
import random
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
CUS, SIMDS, CLOCK_GHZ = 256, 4, 2.4
VALU_SAD_PEAK_GBS = CUS * SIMDS * 32 * 8 * CLOCK_GHZ  # v_sad_u16: 2 sample pairs = 8 algorithmic bytes per lane, 32 lanes per clock and SIMD (a wave64 issues over 2 clocks)
VALU_ISSUE_PEAK_GINST = CUS * SIMDS * CLOCK_GHZ / 2.0  # wave64 VALU instructions per ns: one per SIMD every 2 clocks
BYTES_PER_SEARCH_UNIT = 256  # 64 sample pairs x (2 + 2) bytes (4*w*h per block SAD)
INSTR_PER_BIN = 32  # k_cu_bits: measured instructions per coded bin (counted in the kernel's ISA, 16 bins unrolled)


def reference_noise(nbytes, seed):
    """the byte stream of SURVEY.md 8(d)'s recipe -- random.seed(S); bytes(random.getrandbits(8) ...) -- without the Python loop: getrandbits(8) is the top byte of
    one MT19937 output, and numpy's legacy generator runs the same twister from the same state (checked against the loop in tests/test_bench_inputs.py)"""
    random.seed(seed)
    st = random.getstate()
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.array(st[1][:624], dtype=np.uint32), st[1][624]))
    return (rs.randint(0, 2 ** 32, size=nbytes, dtype=np.uint32) >> 24).astype(np.uint8)


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
    """the reference encoder on this box's host cores: xeveb_app -m 8 and -m 1 side by side on the first `frames` frames of the seed-4 clip (started in the background
    while the GPU encodes; `result()` waits for them)"""

    def __init__(self, width, height, frames, clip, with_m1=False):
        self.w, self.h, self.frames, self.procs, self.err = width, height, frames, {}, None
        self.exe = os.path.join(ROOT, "oracle", "_ref", "xeveb_app")
        if not os.path.exists(self.exe):
            self.err = "oracle/_ref/xeveb_app not built"
            return
        try:
            self.dir = tempfile.mkdtemp(prefix="xeve_bench_")
            yuv = os.path.join(self.dir, "in.yuv")
            clip.tofile(yuv)
            for m in ((8, 1) if with_m1 else (8,)):
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
            tot = re.search(r"Total encoding time\s*=\s*[0-9.]+ msec,\s*([0-9.]+) sec", txt)
            out[m] = {"fps": float(fps.group(1)) if fps else None, "encoding_s": float(tot.group(1)) if tot else None, "rc": p.returncode}
            try:
                out[m]["md5"] = hashlib.md5(open(os.path.join(self.dir, "m%d.evc" % m), "rb").read()).hexdigest()
            except Exception:
                pass
        try:
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)
        except Exception:
            pass
        m8, m1 = out.get(8, {}), out.get(1, {})
        return {"value": m8.get("fps"), "unit": "frames/s", "cores": 8, "kind": "reference",
                "sample": "oracle/_ref/xeveb_app (the unmodified reference, AVX2 dispatch) -w %d -h %d --preset medium --closed-gop -I 8 --frames %d on the seed-4 uniform 8-bit "
                          "4:2:0 clip (frame 0 is the IDR picture, the rest inter pictures) = GOP 0 of the GPU job; `value` = -m 8 (the library's thread maximum, the setting the "
                          "GPU job reproduces byte for byte); `m1` = -m 1 when asked for (--cpu-m1; 0.053 frames/s in profiles/r03_bench.json); on the host's cores while the GPU encoded" % (self.w, self.h, self.frames),
                "m8": m8, "m1": m1, "host": host_info()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gops", type=int, default=0, help="closed GOPs in lockstep per GPU (0: as many as the picture size allows, at most 448)")
    ap.add_argument("--frames", type=int, default=2, help="frames per GOP (2: the IDR picture and one inter picture, the CPU baseline's sample)")
    ap.add_argument("--threads", type=int, default=8, help="row chains per picture = the reference's -m")
    ap.add_argument("--batches", type=int, default=3, help="batches encoded side by side on this GPU (one host thread and HIP stream each): the first of --gops GOPs, the others as the memory allows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-m1", action="store_true", help="time the reference with one thread too (-m 1: ~3 minutes at 3840x2160; profiles/r03_bench.json has it)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed per-kernel-class pass")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # harness self-test only: XEVE_BENCH_SHARE_GPU=1 runs every rank on GPU 0 with the gloo backend, so that the N > 1 control path (barriers, max-over-ranks, rank-0
    # JSON) can be exercised on a one-GPU box; never used for numbers
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
    lib_path = os.path.join(ROOT, "xeve_amd", "lib", "libxeve_hip.so")
    built_here = False
    if not os.path.exists(lib_path) and local == 0:
        import __graft_entry__

        __graft_entry__.build()  # fresh checkout on the GPU box: compile once (hipcc is in the image)
        built_here = True
    if world > 1:
        dist.barrier()
    import xeve_amd
    from xeve_amd import encode, lib

    xeve_amd.init(local)
    solo = rank == 0 and world == 1
    W, H, F, T = a.width, a.height, a.frames, a.threads
    fb = W * H * 3 // 2
    w_lcu, h_lcu = (W + 63) // 64, (H + 63) // 64
    vh = (H + 288 + 63) & ~63
    G = a.gops or max(1, min(448, int((2 ** 32 - 1) // (vh * W))))  # (the library's limit: the stacked originals of a batch below 2^32 samples)
    cfg = encode.config(W, H, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=T)
    # BATCHES: a batch's originals are addressed with 32 bits (448 pictures of 3840x2160), HBM holds more, and a step's time hardly moves with the chains it carries -- so
    # the job is B batches side by side (a host thread and a HIP stream each: profiles/r03s_*): the first as large as the limit allows, the others as large as the
    # memory left over
    free0 = torch.cuda.mem_get_info(dev)[0]
    encs, Gs = [encode.BatchEncoder(cfg, G, F)], [G]
    per_gop = (free0 - torch.cuda.mem_get_info(dev)[0]) / G + (16 << 20)  # (+ the walk's workspace, allocated when the encode begins)
    for _ in range(1, max(1, a.batches)):
        room = torch.cuda.mem_get_info(dev)[0] - (16 << 20) * sum(Gs) - (8 << 30)
        g = int(min(G, room // per_gop))
        if g < max(1, G // 8):
            break
        encs.append(encode.BatchEncoder(cfg, g, F)), Gs.append(g)
    B = len(encs)
    enc = encs[0]

    # inputs: GOP 0 and the last GOP of rank 0 = the reference recipe's seed-4 clip (its bitstream has a golden); every other GOP i.i.d. uniform bytes made on the device
    clip = reference_noise(fb * F, 4) if rank == 0 else None
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    for b, e in enumerate(encs):
        for g in range(Gs[b]):
            if (g == 0 or g == Gs[b] - 1) and clip is not None:  # (both ends of every batch carry the clip: their bytes must be GOP 0's)
                d = torch.from_numpy(clip).to(dev)
            else:
                d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device=dev, generator=gen)
            for f in range(F):
                e.push(g, f, d[f * fb:(f + 1) * fb])
    del d
    cpu = CpuApp(W, H, F, clip, a.cpu_m1) if solo and not a.no_cpu_baseline else None  # (host cores only; runs while the GPU encodes)

    def fence():
        for e in encs:
            e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(len(encs) - 1, -1, -1):  # (the walk's workspace is allocated here: a batch beyond the first that no longer fits is left out rather than failing the run)
        try:
            encs[i].begin()
        except Exception:
            if i == 0:
                raise
            encs[i].close()
            del encs[i], Gs[i]
    B = len(encs)
    total = enc.advance(0)
    per_picture = total // F
    n = a.warmup + a.steps
    per = max(1, total // n)
    sizes = [per] * (n - 1) + [max(0, total - per * (n - 1))]

    live = {"search": [0.0, 0, 0]}

    def drain():
        for k, v in lib.prof_read().items():  # (waits for the device: only between slices; keeps the pool of timing events small -- creating them costs host time)
            if k in live:
                live[k] = [live[k][0] + v[0], live[k][1] + v[1], live[k][2] + v[2]]

    def run_slices(lo, hi, timers=False):
        """slices [lo, hi) of every batch; the batches' host threads issue side by side (the library call releases the GIL).  timers: HIP events on the search kernel's
        launches -- with one batch in every slice (read out after each), with several only in the last slice (a read-out would stall the other batches)"""
        def one(e):
            left = None
            for i in range(lo, hi):
                if timers and (B == 1 or i == hi - 1) and e is enc:
                    lib.prof_enable(["search"])
                left = e.advance(sizes[i])
                if timers and B == 1:
                    drain()
            return left
        if B == 1:
            return [one(enc)]
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(B) as ex:
            return list(ex.map(one, encs))

    run_slices(0, a.warmup)
    fence()
    lib.prof_read()
    fence()
    t0 = time.perf_counter()
    lefts = run_slices(a.warmup, n, timers=True)  # the roofline's kernel: HIP events on its launches, live in the timed region
    fence()
    dt = time.perf_counter() - t0
    drain()
    lib.prof_enable(None)
    assert all(v == 0 for v in lefts), lefts
    timed_steps = sum(sizes[a.warmup:])
    frames_timed = sum(Gs) * timed_steps / per_picture
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = enc.stats()
    streams = [enc.bitstream(0)]
    ends_same = all(e.bitstream(g) == streams[0] for e, n in zip(encs, Gs) for g in (0, n - 1)) if rank == 0 else None
    total_bytes = int(sum(sum(e.bitstream_sizes()) for e in encs)) if rank == 0 else 0
    for e in encs:  # (the job's HBM goes back before the untimed extras)
        e.close()

    line = None
    if rank == 0:
        s_ms, s_n, s_u = live["search"]
        alg = s_u * BYTES_PER_SEARCH_UNIT
        alg_gbs = alg / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
        roof = {"kernel": "k_me_epzs<8|16|32|64, uni|bi> (the integer motion search inside xeve_hip_mode_analyze_ctu_jobs; the SAD kernel of the path)",
                "bound": "valu", "achieved": round(alg_gbs, 1), "peak": VALU_SAD_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / VALU_SAD_PEAK_GBS, 4),
                "how": "algorithmic bytes (256 per 64 sample pairs evaluated, counted on the device) over the kernel's HIP-event time in the timed region, against the rate at "
                       "which the chip's VALUs can issue v_sad_u16 (8 algorithmic bytes per lane and instruction)",
                "launches_in_region": s_n, "avg_launch_ms": round(s_ms / max(1, s_n), 4), "algorithmic_bytes_per_launch": int(alg / max(1, s_n)),
                "hbm_algorithmic_frac": round(alg_gbs / HBM_PEAK_GBS, 4), "traffic": None}
        try:  # physical HBM bytes per launch of the same kernel from the committed PMC passes (separate runs, profiles/)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_search_pmc.json")))
            roof["traffic"] = pmc.get("hbm_bytes_per_launch_x2")
            roof["traffic_is"] = "HBM bytes per launch of the kernel in the PMC run of profiles/r03_search_pmc.json (128 chains in lockstep; this job's launches carry %d)" % (Gs[0] * min(T, h_lcu))
            if pmc.get("hbm_bytes_per_launch_x2") and pmc.get("avg_launch_s"):
                roof["hbm_physical_GBps"] = round(pmc["hbm_bytes_per_launch_x2"] / pmc["avg_launch_s"] / 1e9, 1)
                roof["hbm_physical_frac"] = round(roof["hbm_physical_GBps"] / HBM_PEAK_GBS, 4)
        except Exception:
            pass
        check = {"gop0_md5": hashlib.md5(streams[0]).hexdigest(), "gop0_bytes": len(streams[0]), "total_bytes": total_bytes,
                 "first_and_last_gop_of_every_batch_same_clip_same_bytes": ends_same}
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_v1.json")))
            key = {(3840, 2160, 2, 8): "cfg4_2160p_closedgop_medium_m8", (3840, 2160, 2, 1): "cfg4_2160p_closedgop_medium"}.get((W, H, F, T))
            if key:
                check["reference_golden"] = key
                check["byte_identical_to_the_reference"] = gold[key]["md5"] == check["gop0_md5"]
        except Exception:
            pass
        line = {
            "metric": "encoded frames/sec @ 2160p Baseline medium; SAD-kernel HBM GB/s vs peak",
            "value": round(world * frames_timed / dt, 4),
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
            "value_is": "encoded frames/s of the closed-GOP batch encoder: a real encode to EVC bitstreams (every stage of the reference's xeve_pic on the device, NAL assembly "
                        "on the host), byte-identical to the reference encoder; frames = the timed share of the job's frames",
            "config": {
                "workload": "one encode of %d batches of %s closed GOPs x %d frames per GPU, %dx%d Baseline preset medium (xeveb_app --preset medium --closed-gop -I 8 -m %d semantics), "
                            "i.i.d. uniform 8-bit 4:2:0 input resident in HBM, QP 32; the job's %d lockstep CTU steps cut into %d + %d equal slices"
                            % (B, "+".join(str(g) for g in Gs), F, W, H, T, total, a.warmup, a.steps),
                "batches_side_by_side": B, "gops_in_lockstep": Gs, "frames_per_gop": F, "row_chains_per_picture": T, "chains_in_lockstep": [g * min(T, h_lcu) for g in Gs],
                "lockstep_steps_per_picture": per_picture,
                "lockstep_steps_timed": timed_steps, "frames_in_timed_region": round(frames_timed, 2), "ctus_per_picture": w_lcu * h_lcu,
                "parallelism": "closed-GOP shards per GPU, no collectives" + (" [SELF-TEST: all ranks share GPU 0, gloo]" if share else ""),
                "library_built_in_this_run": built_here,
            },
            "encode": {"host_seconds_issuing_steps": round(stats["step_seconds"], 2), "seconds_in_picture_ends": round(stats["picture_end_seconds"], 2),
                       "note": "whole job (warm-up slices included): time the host spent issuing the lockstep steps, and inside the picture ends (loop filter, second writer "
                               "pass, padding, read-back of the slice data)"},
            "bitstream_check": check,
            "roofline": roof,
        }

    if solo and not a.no_secondary:
        # per kernel class: a short untimed encode (few GOPs, small picture share) with every class's timer on
        try:
            g2 = max(1, G // 4)
            e2 = encode.BatchEncoder(cfg, g2, F)
            d = torch.randint(0, 256, (fb * F,), dtype=torch.uint8, device=dev, generator=gen)
            for g in range(g2):
                for f in range(F):
                    e2.push(g, f, d[f * fb:(f + 1) * fb])
            e2.begin()
            k = 6
            e2.advance(per_picture)       # the IDR picture, untimed and without timers
            e2.sync()
            lib.prof_enable(lib.PROF_CLASSES)
            lib.prof_read()
            e2.advance(k)                 # k steps of the inter picture
            e2.sync()
            allc = lib.prof_read()
            lib.prof_enable(None)
            e2.close()
            kern = {c: {"ms_per_step": round(v[0] / k, 3), "launches_per_step": v[1] // k} for c, v in allc.items()}
            cb = allc["cu_bits"]
            bins_s = cb[2] / (cb[0] * 1e-3) if cb[0] > 0 else 0.0
            kern["cu_bits"].update({"bins_per_step": int(cb[2] / k), "Gbin_per_s": round(bins_s / 1e9, 3)})
            line["kernels"] = kern
            line["kernels"]["note"] = "HIP-event time per lockstep step of the inter picture with %d GOPs (%d chains) in lockstep, all class timers on (untimed extra encode)" % (g2, g2 * min(T, h_lcu))
            tot = sum(v[0] for c, v in allc.items() if c != "cu_bits_slow")
            line["roofline"]["by_time"] = {
                "kernel": "k_cu_bits (CABAC bit counting, one lane per job): the class with the largest share of the GPU time", "share_of_timed_classes": round(cb[0] / tot, 3) if tot else None,
                "bound": "valu-issue", "achieved": round(bins_s * INSTR_PER_BIN / 64 / 1e9, 3), "peak": round(VALU_ISSUE_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                "frac": round(bins_s * INSTR_PER_BIN / 64 / 1e9 / VALU_ISSUE_PEAK_GINST, 6),
                "how": "bins/s x %d instructions per bin (measured, profiles/) / 64 lanes, against one wave64 VALU instruction per SIMD every 2 clocks on %d CUs x %d SIMDs at %.1f GHz; "
                       "a serial chain per lane, so the roof is only reachable with every lane of every wave busy" % (INSTR_PER_BIN, CUS, SIMDS, CLOCK_GHZ)}
        except Exception as e:  # noqa: BLE001
            line["kernels"] = {"error": repr(e)[:300]}
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu.result()
            try:
                m8 = line["cpu_baseline"].get("m8", {})
                if m8.get("md5") and F == 2 and T == 8:
                    line["bitstream_check"]["same_as_this_runs_cpu_reference_m8"] = m8["md5"] == line["bitstream_check"]["gop0_md5"]
            except Exception:
                pass
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
