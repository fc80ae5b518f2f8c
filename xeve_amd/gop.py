"""Closed-GOP sharding of a sequence over GPUs / encoder processes.

With ``--closed-gop -I K`` every K-frame group starts with an IDR and is encoded independently of all
others (reference: src_base/xeve_enc.c:1083-1095,1154-1167,1980-1993); concatenating the per-group
bitstreams in order is byte-identical to the monolithic encode.  So the multi-GPU path needs NO
collective: rank r of `world` takes GOPs r, r + world, r + 2*world, ... and the host concatenates.
"""
from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class Shard:
    gop: int  # GOP index in display order
    seek: int  # first input frame       (xeve_app --seek, app/xeve_app.c:1171-1202)
    frames: int  # number of input frames  (xeve_app --frames)


def plan(total_frames: int, keyint: int) -> List[Shard]:
    if total_frames < 0 or keyint <= 0:
        raise ValueError("total_frames >= 0 and keyint > 0 required")
    return [Shard(g, s, min(keyint, total_frames - s)) for g, s in enumerate(range(0, total_frames, keyint))]


def shards_for_rank(total_frames: int, keyint: int, rank: int, world: int) -> List[Shard]:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [s for s in plan(total_frames, keyint) if s.gop % world == rank]


def app_args(shard: Shard) -> List[str]:
    """CLI fragment that makes the reference app encode exactly this shard."""
    return ["--seek", str(shard.seek), "--frames", str(shard.frames)]


def concat_order(per_rank: List[List[Shard]]) -> List[Shard]:
    """Order in which the host concatenates the shard bitstreams gathered from all ranks."""
    allsh = sorted((s for lst in per_rank for s in lst), key=lambda s: s.gop)
    if [s.gop for s in allsh] != list(range(len(allsh))):
        raise ValueError("shards do not tile the sequence exactly once")
    return allsh


# ---------------------------------------------------------------------------------------------------------------------------
# the shard driver: one encoder process per GPU, each encoding its closed GOPs, the host concatenating the bitstreams
# ---------------------------------------------------------------------------------------------------------------------------
def run_shards(encoder_cmd, yuv, out, total_frames, keyint, devices, env=None, work_dir=None, per_device=1, timeout=None):
    """Encode `yuv` as independent closed GOPs, one encoder PROCESS per entry of `devices` (several GOPs of a device one after the other,
    `per_device` processes side by side on each), and concatenate the shard bitstreams in GOP order into `out`.

    encoder_cmd : argv of the encoder up to the shard-specific options, e.g. [xeveb_app, "-i", yuv, "-w", "3840", "-h", "2160", "--preset",
                  "medium", "--closed-gop", "-I", "8", "-m", "1"] -- "--seek S --frames K -o PART" are appended per shard.  The encoder is the
                  reference's own (the host code stays C, BASELINE.json north_star); the GPU comes in through the environment: `env` is
                  merged into every process's environment (e.g. LD_PRELOAD of the integration shim, XEVE_HIP_LIB, XEVE_HIP_SHIM_*) and the
                  process for device d additionally gets HIP_VISIBLE_DEVICES=d and XEVE_HIP_DEVICE=0, so every process sees exactly ONE GPU:
                  the process-wide dispatch tables of the reference (src_base/xeve_sad.c:34-37) bind one process to one GPU.
    There is no exchange between the processes at all (no RCCL, nothing over xGMI): the only joint operation is the concatenation.
    Returns {"bytes": n, "shards": [(gop, device, seconds)], "seconds": wall, "fps": frames / wall}."""
    import os
    import shutil
    import subprocess
    import tempfile
    import time

    shards = plan(total_frames, keyint)
    if not shards:
        raise ValueError("nothing to encode")
    slots = [d for d in devices for _ in range(per_device)]
    if not slots:
        raise ValueError("no device given")
    own_dir = work_dir is None
    work_dir = work_dir or tempfile.mkdtemp(prefix="xeve_shards_")
    queues = {i: [s for s in shards if s.gop % len(slots) == i] for i in range(len(slots))}  # GOP g -> slot g mod n, as shards_for_rank
    running, done, t0 = {}, [], time.perf_counter()

    def start(i):
        if not queues[i]:
            return
        s = queues[i].pop(0)
        part = os.path.join(work_dir, "gop%06d.evc" % s.gop)
        e = dict(os.environ)
        e.update(env or {})
        e["HIP_VISIBLE_DEVICES"], e["XEVE_HIP_DEVICE"] = str(slots[i]), "0"
        log = open(part + ".stderr", "w+")  # (a file, not a pipe: a chatty child must never block on a full pipe buffer while we only poll)
        p = subprocess.Popen(list(encoder_cmd) + app_args(s) + ["-o", part], env=e, stdout=subprocess.DEVNULL, stderr=log, text=True)
        running[i] = (p, s, part, time.perf_counter(), log)

    try:
        for i in range(len(slots)):
            start(i)
        while running:
            for i in list(running):
                p, s, part, ts, log = running[i]
                if p.poll() is None:
                    if timeout and time.perf_counter() - ts > timeout:  # per shard
                        raise TimeoutError("the encode of GOP %d exceeded %s s" % (s.gop, timeout))
                    continue
                log.seek(0)
                err = log.read()
                log.close()
                if p.returncode != 0 or not os.path.exists(part):
                    raise RuntimeError("encoder failed on GOP %d (device %s, rc %s): %s" % (s.gop, slots[i], p.returncode, err[-800:]))
                done.append((s.gop, slots[i], time.perf_counter() - ts, part))
                del running[i]
                start(i)
            time.sleep(0.02)
        wall = time.perf_counter() - t0
        done.sort()
        if [g for g, _, _, _ in done] != list(range(len(shards))):
            raise RuntimeError("shards do not tile the sequence exactly once")
        n = 0
        with open(out, "wb") as f:
            for _, _, _, part in done:
                with open(part, "rb") as g:
                    n += f.write(g.read())
        return {"bytes": n, "shards": [(g, d, round(t, 3)) for g, d, t, _ in done], "seconds": wall, "fps": total_frames / wall}
    finally:
        for p, _, _, _, log in running.values():
            p.kill()
            p.wait()
            log.close()
        if own_dir:
            shutil.rmtree(work_dir, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------------
# the same on the batch encoder (xeve_amd/encode.py): one PROCESS per GPU, each coding ITS closed GOPs in lockstep batches
# ---------------------------------------------------------------------------------------------------------------------------
def run_encoder_shards(yuv, out, config, total_frames, keyint, devices, per_device=1, work_dir=None, timeout=None, worker_cmd=None, env=None, aligned_only=True):
    """Encode `yuv` as closed GOPs of `keyint` frames on the batch encoder, one worker PROCESS per entry of `devices` (x per_device): worker i of n codes the GOPs
    g = i, i + n, i + 2n, ... -- all of them at once, in lockstep batches side by side (encode.encode_gops) -- and the host concatenates the per-GOP bitstreams in order.

    config     : the keyword arguments of xeve_amd.encode.config (w, h, qp, bframes, preset, threads, input_depth, ...); closed_gop and keyint are set here.
    worker_cmd : argv prefix of the worker (default: this interpreter running xeve_amd.shard_worker); it gets one argument, the path of its job file (JSON).
    No exchange between the workers (no RCCL, nothing over xGMI): the only joint operation is the concatenation.
    The joined file is the reference's ONE run over the sequence when keyint is a multiple of bframes + 1 (or bframes is 0): with a keyint that cuts a sub-GOP short the
    reference itself codes a GOP's tail differently inside a sequence than as a run of its own (its --seek / --frames runs do not concatenate to its single run either;
    found with tests/golden/fuzz_enc_host.py-style runs), so such a split is refused unless aligned_only=False -- every part is then still the reference's run over that GOP.
    Returns {"bytes": n, "workers": [(worker, device, seconds)], "seconds": wall, "fps": frames / wall}."""
    import json
    import os
    import shutil
    import subprocess
    import sys
    import tempfile
    import time

    shards = plan(total_frames, keyint)
    if not shards:
        raise ValueError("nothing to encode")
    bframes = int(config.get("bframes", 15))
    if aligned_only and bframes and keyint % (bframes + 1) != 0:
        raise ValueError("keyint %d is not a multiple of bframes + 1 = %d: the parts would not join to the reference's single run (aligned_only=False to split anyway)" % (keyint, bframes + 1))
    slots = [d for d in devices for _ in range(per_device)]
    if not slots:
        raise ValueError("no device given")
    own_dir = work_dir is None
    work_dir = work_dir or tempfile.mkdtemp(prefix="xeve_shards_")
    cfg = dict(config)
    cfg.update(closed_gop=True, keyint=keyint)
    procs, t0 = [], time.perf_counter()
    try:
        for i, d in enumerate(slots):
            if not any(s.gop % len(slots) == i for s in shards):
                continue
            job = os.path.join(work_dir, "worker%03d.json" % i)
            with open(job, "w") as f:
                json.dump({"yuv": os.path.abspath(yuv), "config": cfg, "total_frames": total_frames, "keyint": keyint, "rank": i, "world": len(slots), "dir": work_dir,
                           "memory_share": 1.0 / per_device}, f)  # (workers sharing a GPU plan their batches on their share of its free memory)
            e = dict(os.environ)
            e.update(env or {})
            e["HIP_VISIBLE_DEVICES"], e["XEVE_HIP_DEVICE"] = str(d), "0"  # (every worker sees exactly one GPU)
            log = open(job + ".stderr", "w+")
            p = subprocess.Popen(list(worker_cmd or [sys.executable, "-m", "xeve_amd.shard_worker"]) + [job], env=e, stdout=subprocess.DEVNULL, stderr=log, text=True)
            procs.append([p, i, d, time.perf_counter(), log, None])
        while any(r[5] is None for r in procs):
            for r in procs:
                p, i, d, ts, log, secs = r
                if secs is not None:
                    continue
                if p.poll() is None:
                    if timeout and time.perf_counter() - ts > timeout:
                        raise TimeoutError("worker %d (device %s) exceeded %s s" % (i, d, timeout))
                    continue
                log.seek(0)
                err = log.read()
                if p.returncode != 0:
                    raise RuntimeError("worker %d failed (device %s, rc %s): %s" % (i, d, p.returncode, err[-800:]))
                r[5] = time.perf_counter() - ts
            time.sleep(0.02)
        wall, n = time.perf_counter() - t0, 0
        with open(out, "wb") as f:
            for s in shards:
                part = os.path.join(work_dir, "gop%06d.evc" % s.gop)
                if not os.path.exists(part):
                    raise RuntimeError("no bitstream for GOP %d" % s.gop)
                with open(part, "rb") as g:
                    n += f.write(g.read())
        return {"bytes": n, "workers": [(i, d, round(secs, 3)) for _, i, d, _, _, secs in procs], "seconds": wall, "fps": total_frames / wall}
    finally:
        for p, _, _, _, log, _ in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
            log.close()
        if own_dir:
            shutil.rmtree(work_dir, ignore_errors=True)
