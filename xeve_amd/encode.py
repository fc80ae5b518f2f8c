"""The closed-GOP batch encoder (include/xeve_hip.h: xeve_hip_enc_*; xeve_amd/csrc/encode.hip + enc_host.h): G independent runs of F frames each -- one closed GOP
per run, coded as the reference application codes `--seek g*F --frames F` -- advance in lockstep on the device; every run's bitstream is byte-identical to the
reference's.  This module is the ctypes face of it plus the file plumbing (YUV in, .evc out).  No CPU path: it raises when the library or the GPU is missing."""
import ctypes as C

from . import lib as _lib

PRESETS = {"fast": 0, "medium": 1, "slow": 2, "placebo": 3}


def config(w, h, qp=32, keyint=0, bframes=15, closed_gop=False, preset="medium", threads=1, fps=(30, 1), ref=0, always_second_pass=False, input_depth=8, level_idc=40, sei_info=True,
           inter_slice_type=0, qp_cb_offset=0, qp_cr_offset=0):
    """the options of the reference application (xeve_app: -w -h -q -I -b --closed-gop --preset -m -z --ref -d; --inter-slice-type 0 B / 1 P, --qp-cb-offset,
    --qp-cr-offset as the reference LIBRARY takes them) as the library's configuration record"""
    c = _lib.EncConfig()
    c.w, c.h, c.fps_num, c.fps_den, c.qp, c.keyint, c.bframes, c.closed_gop = w, h, fps[0], fps[1], qp, keyint, bframes, int(bool(closed_gop))
    c.preset, c.threads, c.inter_slice_type, c.ref = PRESETS[preset] if isinstance(preset, str) else int(preset), threads, int(inter_slice_type), ref
    c.reserved[2], c.reserved[3] = int(qp_cb_offset), int(qp_cr_offset)
    c.reserved[0] = (1 if always_second_pass else 0) | (0 if sei_info else 2) | ((int(level_idc) & 0xFF) << 8)  # (--info 0, --level-idc)
    c.reserved[1] = int(input_depth)
    return c


class BatchEncoder:
    """enc = BatchEncoder(cfg, ngops, frames); enc.push(g, f, frame_bytes | device tensor) ...; streams = enc.encode()"""

    def __init__(self, cfg, ngops, frames):
        L = _lib.load()
        self._L, self.cfg, self.ngops, self.frames = L, cfg, ngops, frames
        self.frame_bytes = cfg.w * cfg.h * 3 // 2 * (2 if cfg.reserved[1] > 8 else 1)
        self._h = L.xeve_hip_enc_create(C.byref(cfg), ngops, frames)
        if not self._h:
            raise _lib.XeveHipError(_lib.last_error())

    def close(self):
        if self._h:
            self._L.xeve_hip_enc_delete(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def push(self, gop, frame, data):
        """data: bytes-like of one planar 4:2:0 frame (host; one byte per sample, two little-endian with input_depth 10), or a uint8 torch tensor of those bytes on the GPU"""
        if hasattr(data, "data_ptr"):
            assert data.numel() == self.frame_bytes and data.is_contiguous()
            if data.is_cuda:  # the library copies on its own stream: whatever produced the tensor on torch's stream must be through
                import torch

                torch.cuda.current_stream(data.device).synchronize()
            _lib.check(self._L.xeve_hip_enc_push(self._h, gop, frame, C.c_void_p(data.data_ptr()), 1 if data.is_cuda else 0))
        else:
            b = bytes(data)
            assert len(b) == self.frame_bytes
            _lib.check(self._L.xeve_hip_enc_push(self._h, gop, frame, b, 0))

    def push_gop(self, gop, data):
        for f in range(self.frames):
            self.push(gop, f, data[f * self.frame_bytes:(f + 1) * self.frame_bytes])

    def begin(self):
        _lib.check(self._L.xeve_hip_enc_begin(self._h))

    def advance(self, max_steps):
        """issues up to max_steps lockstep steps (one CTU of every row chain of every GOP each); returns the steps still to do"""
        left = C.c_int64()
        _lib.check(self._L.xeve_hip_enc_advance(self._h, int(max_steps), C.byref(left)))
        return left.value

    def sync(self):
        _lib.check(self._L.xeve_hip_enc_sync(self._h))

    def flush(self):
        """appends the access unit still outstanding (a picture's is appended when the next picture ends): bitstream() then holds every picture whose steps were issued"""
        _lib.check(self._L.xeve_hip_enc_flush(self._h))

    def bitstream(self, gop):
        """the bitstream of one GOP (bytes)"""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self._L.xeve_hip_enc_bitstream(self._h, gop, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value) if n.value else b""

    def bitstream_sizes(self):
        out = []
        for g in range(self.ngops):
            p, n = C.c_void_p(), C.c_size_t()
            _lib.check(self._L.xeve_hip_enc_bitstream(self._h, g, C.byref(p), C.byref(n)))
            out.append(int(n.value))
        return out

    def bitstreams(self):
        return [self.bitstream(g) for g in range(self.ngops)]

    def encode(self):
        """codes every run; returns the list of bitstreams (bytes), one per GOP"""
        _lib.check(self._L.xeve_hip_enc_encode(self._h))
        out = []
        for g in range(self.ngops):
            p, n = C.c_void_p(), C.c_size_t()
            _lib.check(self._L.xeve_hip_enc_bitstream(self._h, g, C.byref(p), C.byref(n)))
            out.append(C.string_at(p.value, n.value) if n.value else b"")
        return out

    def stats(self):
        steps, a, b = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self._L.xeve_hip_enc_stats(self._h, C.byref(steps), C.byref(a), C.byref(b)))
        return {"ctu_steps": steps.value, "step_seconds": a.value, "picture_end_seconds": b.value}


class walk_select:
    """with walk_select(0 | 1 | -1, chains_per_team=0, side=-1): ... -- pins the composed walk (0) / the fused kernel (1) / the choice by width (-1) for the encoders created
    inside, optionally how many chains a team of the fused kernel carries, and whether the composed walk uses its side stream (1 / 0; -1 leaves it as it is)
    (xeve_hip_walk_select / _team / _side: process-wide; create AND close the encoder inside)"""

    def __init__(self, mode, chains_per_team=0, side=-1):
        self.mode, self.team, self.side = int(mode), int(chains_per_team), int(side)

    def __enter__(self):
        L = _lib.load()
        self.before, self.team_before, self.side_before = L.xeve_hip_walk_select(self.mode), L.xeve_hip_walk_team(self.team), L.xeve_hip_walk_side(self.side)
        return self

    def __exit__(self, *exc):
        L = _lib.load()
        L.xeve_hip_walk_select(self.before), L.xeve_hip_walk_team(self.team_before), L.xeve_hip_walk_side(self.side_before)
        return False


def footprint(cfg, ngops, frames):
    """(device bytes a batch of ngops x frames takes, the most GOPs one batch can hold at this picture size) -- xeve_hip_enc_footprint; no device call"""
    L = _lib.load()
    b, m = C.c_uint64(), C.c_int32()
    _lib.check(L.xeve_hip_enc_footprint(C.byref(cfg), int(ngops), int(frames), C.byref(b), C.byref(m)))
    return int(b.value), int(m.value)


def plan_batches(cfg, ngops, frames, free_bytes, max_batches=3, reserve_bytes=8 << 30, batch_gops=None):
    """How `ngops` GOPs are cut into batches that run side by side on one GPU (a host thread and a stream each): as few batches as the per-batch limit (32-bit offsets into
    the stacked originals) and the free HBM allow, at most max_batches AT A TIME -- what is left over waits for the next round.  Returns a list of rounds, each a list of
    (first GOP, GOPs).  A step's time hardly depends on the chains it carries and the device runs the batches' launch chains side by side, so frames/s grows with the
    GOPs in flight (profiles/r03t_*): a round is filled as far as the memory goes.  batch_gops: a lower cap on a batch than the library's (tests)."""
    one, most = footprint(cfg, 1, frames)
    if batch_gops:
        most = min(most, int(batch_gops))
    two, _ = footprint(cfg, 2, frames)
    per_gop, fixed = max(1, two - one), max(0, 2 * one - two)
    rounds, at = [], 0
    while at < ngops:
        room, batches = free_bytes - reserve_bytes, []
        while at < ngops and len(batches) < max_batches:
            g = int(min(most, ngops - at, (room - fixed) // per_gop))
            if g < 1:
                break
            batches.append((at, g))
            at += g
            room -= fixed + g * per_gop
        if not batches:
            raise _lib.XeveHipError("not enough device memory for a single GOP of this size")
        rounds.append(batches)
    return rounds


def encode_gops(cfg, ngops, frames, feed, free_bytes=None, max_batches=3, batch_gops=None):
    """codes `ngops` closed GOPs of `frames` pictures each on the current GPU and returns their bitstreams in order.  feed(encoder, first_gop, count) pushes the frames of
    GOPs [first_gop, first_gop + count) into `encoder` as its GOPs 0 .. count - 1.  The GOPs are cut into batches by plan_batches; the batches of a round are encoded
    side by side, one host thread each (the library call releases the GIL)."""
    import concurrent.futures as cf

    if free_bytes is None:
        import torch

        free_bytes = torch.cuda.mem_get_info()[0]
    out = [None] * ngops
    for batches in plan_batches(cfg, ngops, frames, free_bytes, max_batches, batch_gops=batch_gops):
        encs = [BatchEncoder(cfg, n, frames) for _, n in batches]
        try:
            for e, (first, n) in zip(encs, batches):
                feed(e, first, n)
            with cf.ThreadPoolExecutor(len(encs)) as ex:
                results = list(ex.map(lambda e: e.encode(), encs))
            for (first, n), streams in zip(batches, results):
                out[first:first + n] = streams
        finally:
            for e in encs:
                e.close()
    return out


def encode_file(yuv_path, out_path, cfg, gops, frames, max_batches=3):
    """the whole sequence: GOP g = frames [g * frames, (g + 1) * frames) of the file; the concatenated bitstreams are what the reference writes for the same sequence
    with --closed-gop -I frames.  Sequences beyond one batch are cut into batches that run side by side (encode_gops)."""
    fb = cfg.w * cfg.h * 3 // 2 * (2 if cfg.reserved[1] > 8 else 1)

    def feed(enc, first, n):
        with open(yuv_path, "rb") as f:
            f.seek(first * frames * fb)
            for g in range(n):
                for k in range(frames):
                    enc.push(g, k, f.read(fb))

    streams = encode_gops(cfg, gops, frames, feed, max_batches=max_batches)
    with open(out_path, "wb") as f:
        for s in streams:
            f.write(s)
    return streams
