"""Helpers of the batch-encoder tests: the configuration record of include/xeve_hip.h, the CPU harness (oracle/libxeve_enc_oracle.so = the product's frame loop
on the oracle's engine; test infrastructure), the command-line <-> configuration mapping, and the cases whose goldens tests/golden/make_enc_golden.py records."""
import ctypes as C
import hashlib
import json
import os
import subprocess

from _libs import ORACLE_DIR, ROOT

ENC_ORACLE_SO = os.path.join(ORACLE_DIR, "libxeve_enc_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "enc_v1.json")


class EncConfig(C.Structure):  # xeve_hip_enc_config
    _fields_ = [(n, C.c_int32) for n in "w h fps_num fps_den qp keyint bframes closed_gop preset threads inter_slice_type ref".split()] + [("reserved", C.c_int32 * 4)]


def config(w, h, cli, threads=1):
    """the application's options -> the configuration record (defaults: xeve_param_init, xeve_enc.c:2290-2324)"""
    c = EncConfig()
    c.w, c.h, c.fps_num, c.fps_den, c.qp, c.bframes, c.preset, c.threads = w, h, 30, 1, 32, 15, 1, threads
    i = 0
    while i < len(cli):
        a = cli[i]
        if a == "--preset":
            c.preset = {"fast": 0, "medium": 1, "slow": 2, "placebo": 3}[cli[i + 1]]
        elif a == "-I":
            c.keyint = int(cli[i + 1])
        elif a == "-b":
            c.bframes = int(cli[i + 1])
        elif a == "--ref":
            c.ref = int(cli[i + 1])
        elif a == "-q":
            c.qp = int(cli[i + 1])
        elif a == "-m":
            c.threads = int(cli[i + 1])
        elif a == "-d":
            c.reserved[1] = int(cli[i + 1])
        elif a == "--info":
            c.reserved[0] = (c.reserved[0] & ~2) | (0 if int(cli[i + 1]) else 2)
        elif a == "--level-idc":
            c.reserved[0] = (c.reserved[0] & ~0xFF00) | (int(cli[i + 1]) << 8)
        elif a == "--inter-slice-type":
            c.inter_slice_type = int(cli[i + 1])
        elif a == "--qp-cb-offset":
            c.reserved[2] = int(cli[i + 1])
        elif a == "--qp-cr-offset":
            c.reserved[3] = int(cli[i + 1])
        elif a == "--closed-gop":
            c.closed_gop = 1
            i -= 1
        else:
            raise ValueError(a)
        i += 2
    return c


_harness = None


def harness():
    global _harness
    if _harness is None and os.environ.get("XO_ENC_ORACLE_LIB"):  # (tests/test_walk_race.py: the ThreadSanitizer build of the same harness)
        _harness = C.CDLL(os.environ["XO_ENC_ORACLE_LIB"])
    if _harness is None:
        # make decides (oracle/Makefile lists enc_oracle.cpp, the oracle, enc_host.h / enc_plan.h AND every walk*.h the harness compiles: an edit to the walk alone
        # must not be tested against a stale build)
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "enc"])
        _harness = C.CDLL(ENC_ORACLE_SO)
    return _harness


def plan_cpu(cfg, frames):
    """[(frame, poc, slice type, tid, qp, idr, L0 poc, L1 poc)] of the coded pictures, from the product's frame loop"""
    buf = (C.c_int32 * (8 * frames))()
    n = harness().xo_encode_plan(C.byref(cfg), frames, buf, frames)
    assert n == frames, n
    return [tuple(buf[8 * i:8 * i + 8]) for i in range(n)]


def encode_cpu(cfg, gops, frames, always_rewrite=False):
    """gops: list of bytes objects (planar 8-bit 4:2:0 frames of one run each) -> list of bitstreams"""
    G = len(gops)
    arr = (C.c_char_p * G)(*gops)
    out, nb, err = (C.POINTER(C.c_uint8) * G)(), (C.c_size_t * G)(), C.create_string_buffer(512)
    rc = harness().xo_encode_gops(C.byref(cfg), C.cast(arr, C.POINTER(C.c_void_p)), G, frames, int(always_rewrite), out, nb, err, 512)
    if rc:
        raise RuntimeError(err.value.decode())
    res = [bytes(bytearray(out[g][:nb[g]])) for g in range(G)]
    for g in range(G):
        harness().xo_encode_free(out[g])
    return res


def encode_cpu_flushed(cfg, gops, frames, pictures_per_slice=1):
    """the same run cut after every `pictures_per_slice` pictures with the frame loop's flush() at each cut -> (bitstreams, [bytes of GOP 0's stream after each cut])"""
    G = len(gops)
    arr = (C.c_char_p * G)(*gops)
    out, nb, err = (C.POINTER(C.c_uint8) * G)(), (C.c_size_t * G)(), C.create_string_buffer(512)
    after, ncuts = (C.c_size_t * (frames + 2))(), C.c_int()
    rc = harness().xo_encode_gops_flushed(C.byref(cfg), C.cast(arr, C.POINTER(C.c_void_p)), G, frames, int(pictures_per_slice), out, nb, after, frames + 2, C.byref(ncuts), err, 512)
    if rc:
        raise RuntimeError(err.value.decode())
    res = [bytes(bytearray(out[g][:nb[g]])) for g in range(G)]
    for g in range(G):
        harness().xo_encode_free(out[g])
    return res, [int(after[i]) for i in range(ncuts.value)]


def md5(b):
    return hashlib.md5(b).hexdigest()


# Frame-loop grid: (options, frame counts) whose picture tables (POC order, temporal id, type, QP, first reference of each list) the golden file records from the
# reference application's own report
PLAN_GRID = [
    (["--preset", "fast", "-I", "0", "-b", "0"], (1, 2, 5, 9, 17)),
    (["--preset", "fast", "-I", "5", "-b", "0"], (3, 8, 12)),
    (["--preset", "fast", "-I", "6", "-b", "0", "--closed-gop"], (5, 13)),
    (["--preset", "fast", "-I", "1", "-b", "0"], (3,)),
    (["--preset", "fast", "-b", "1"], (2, 3, 8, 9)),
    (["--preset", "medium"], (1, 2, 3, 5, 8, 9, 17, 20, 33)),
    (["--preset", "medium", "-b", "7"], (3, 8, 9, 17, 20)),
    (["--preset", "medium", "-b", "3", "-I", "8"], (5, 8, 9, 17, 20)),
    (["--preset", "medium", "-b", "15", "-I", "32"], (17, 33, 40)),
    (["--preset", "medium", "--closed-gop", "-I", "8"], (1, 2, 3, 4, 5, 6, 7, 8, 9, 17, 20)),
    (["--preset", "medium", "--closed-gop", "-I", "8", "-b", "3"], (5, 8, 20)),
    (["--preset", "medium", "--closed-gop", "-I", "4", "-b", "7"], (5, 9, 17)),
    (["--preset", "medium", "--closed-gop", "-I", "12", "-b", "7"], (9, 20, 33)),
]

# Batches of closed GOPs: name -> (w, h, gops, frames per gop, seed, options, threads).  The golden holds one md5 per GOP, from the reference application run with
# --seek g * frames --frames frames.  Seeds as in tests/_e2e.py make_yuv (>= 5000: drifting texture).
BATCH_CASES = {
    "gops_128x64_noise": (128, 64, 3, 8, 21, ["--preset", "medium", "--closed-gop", "-I", "8"], 1),
    "gops_128x128_moving_m2": (128, 128, 2, 8, 5021, ["--preset", "medium", "--closed-gop", "-I", "8"], 2),
    "gops_192x256_noise_m3": (192, 256, 2, 4, 23, ["--preset", "fast", "--closed-gop", "-I", "4", "-b", "3"], 3),
    "gops_cif_moving": (352, 288, 4, 8, 5024, ["--preset", "medium", "--closed-gop", "-I", "8"], 1),  # G >= 4 GOPs x 8 frames at 352x288
    "gops_cif_noise_m8": (352, 288, 2, 8, 25, ["--preset", "medium", "--closed-gop", "-I", "8"], 8),
}
# the same at BASELINE's picture sizes (GPU suite only; the CPU harness would need minutes): >= 2 GOPs at 1920x1080
BATCH_CASES_REAL = {
    "gops_1080p_moving_m8": (1920, 1080, 2, 8, 5026, ["--preset", "medium", "--closed-gop", "-I", "8"], 8),
    # BASELINE.json's configs 2, 3, 4 as single runs through the batch encoder: 720p low-delay B with 8 frames, 1080p random access with 9 frames
    # (two B layers below the key pictures, reference distances 8 / 4 / 2 / 1), 2160p closed GOP (the IDR picture and one inter picture; the input of
    # tests/golden/e2e_v1.json cfg4_2160p_closedgop_medium_m8, whose md5 this golden repeats)
    "cfg2_720p_ldb_fast_8f_m8": (1280, 720, 1, 8, 2, ["--preset", "fast", "-b", "0", "-I", "0"], 8),
    "cfg3_1080p_ra_medium_9f_m8": (1920, 1080, 1, 9, 3, ["--preset", "medium"], 8),
    "cfg4_2160p_closedgop_medium_2f_m8": (3840, 2160, 1, 2, 4, ["--preset", "medium", "--closed-gop", "-I", "8"], 8),
    # configs 2 and 3 over a whole 16-picture sub-GOP and the key picture after it (17 frames: four B layers, reference distances 16 / 8 / 4 / 2 / 1 in
    # config 3; sixteen low-delay B pictures whose reference lists slide over the decoded picture buffer in config 2)
    "cfg2_720p_ldb_fast_17f_m8": (1280, 720, 1, 17, 2, ["--preset", "fast", "-b", "0", "-I", "0"], 8),
    "cfg3_1080p_ra_medium_17f_m8": (1920, 1080, 1, 17, 3, ["--preset", "medium"], 8),
    # VERDICT r05 "missing" 4: config 2 at its STATED length -- one 64-frame low-delay GOP at 1280x720 (63 B pictures behind the I picture: the reference lists slide over the
    # decoded picture buffer for four sub-GOPs, the QP offsets of the low-delay hierarchy repeat every 8 pictures)
    "cfg2_720p_ldb_fast_64f_m8": (1280, 720, 1, 64, 2, ["--preset", "fast", "-b", "0", "-I", "0"], 8),
}

# 10-bit input (the application's -d 10: 16-bit little-endian samples, handed to the codec as they are): the 8-bit clip of the seed widened by widen10
DEPTH10_CASES = {
    "gops_128x64_10bit_m2": (128, 64, 2, 8, 27, ["--preset", "medium", "--closed-gop", "-I", "8", "-d", "10"], 2),
    "gops_192x128_10bit_b3": (192, 128, 2, 4, 5028, ["--preset", "fast", "--closed-gop", "-I", "4", "-b", "3", "-d", "10"], 1),
}

# Options the reference APPLICATION lists and fails to parse (--inter-slice-type, --qp-cb-offset, --qp-cr-offset): the goldens come from the reference library with the
# parameter set on the way into xeve_create (oracle/ref_param_pin.c, LD_PRELOAD).  Host side only: the product refuses these until the device path has coded them.
HOST_PINNED_CASES = {
    "p_slices_ldb_9f": (128, 64, 1, 9, 5031, ["--preset", "fast", "-b", "0", "--inter-slice-type", "1"], 1),
    "p_slices_ldb_ref3_m2": (136, 72, 1, 8, 6032, ["--preset", "medium", "-b", "0", "--ref", "3", "--inter-slice-type", "1"], 2),
    "p_slices_hierarchical_closed": (128, 128, 2, 8, 5033, ["--preset", "medium", "--closed-gop", "-I", "8", "-b", "7", "--inter-slice-type", "1"], 2),
    "chroma_qp_offsets": (136, 72, 2, 4, 29, ["--preset", "medium", "--closed-gop", "-I", "4", "-b", "3", "--qp-cb-offset", "5", "--qp-cr-offset", "-6"], 1),
    "chroma_qp_offsets_low_qp_p": (128, 64, 1, 6, 5034, ["--preset", "fast", "-b", "0", "-q", "14", "--qp-cb-offset", "-12", "--qp-cr-offset", "12", "--inter-slice-type", "1"], 1),
}
# header-only options (--info 0: no SEI with the option list; --level-idc): nothing on the device changes
HEADER_OPTION_CASES = {
    "no_info_sei_level_51": (128, 64, 2, 4, 5035, ["--preset", "medium", "--closed-gop", "-I", "4", "-b", "3", "--info", "0", "--level-idc", "51"], 1),
    "level_62_m2": (136, 72, 1, 5, 37, ["--preset", "fast", "-b", "0", "--level-idc", "62"], 2),
}
# --preset slow (quarter-pel search, ME range 128, rdo_dbk_switch = 1: walk_dbk.h / xo_delta_dist) as batches of closed GOPs: several GOPs in lockstep, 3 and 8 row chains
# (a CU at the top of a CTU row filters against the row chain above it: written CTUs, whose luma cbf flags the writer has set)
SLOW_BATCH_CASES = {
    "slow_gops_192x128_moving_m3": (192, 128, 3, 4, 5041, ["--preset", "slow", "--closed-gop", "-I", "4", "-b", "3"], 3),
    "slow_gops_256x192_noise_m8": (256, 192, 2, 2, 43, ["--preset", "slow", "--closed-gop", "-I", "8"], 8),
}
# --preset placebo the same way (4x4 inter CUs, two reference pictures per list, raster search, four merge candidates)
PLACEBO_BATCH_CASES = {
    "placebo_gops_192x128_moving_m3": (192, 128, 3, 4, 5047, ["--preset", "placebo", "--closed-gop", "-I", "4", "-b", "3"], 3),
    "placebo_gops_256x192_noise_m8": (256, 192, 2, 2, 48, ["--preset", "placebo", "--closed-gop", "-I", "8"], 8),
}
# presets slow and placebo at BASELINE's 1920x1080 with the reference's maximum of 8 row chains (GPU suite only): noise for slow, the drifting texture for placebo
PRESET_REAL_CASES = {
    "slow_1080p_noise_3f_m8": (1920, 1080, 1, 3, 3, ["--preset", "slow", "--closed-gop", "-I", "8"], 8),
    "placebo_1080p_moving_3f_m8": (1920, 1080, 1, 3, 5049, ["--preset", "placebo", "--closed-gop", "-I", "8"], 8),
}
# What the PRODUCT library's configuration check (enc_plan.h Param::finish) takes and what it refuses -- ONE table, read by tests/test_enc_host.py through
# xeve_hip_enc_footprint (no device: a change of the accepted set fails in the build container) and by tests/test_enc_gpu.py through xeve_hip_enc_create.
# (kwargs of xeve_amd.encode.config beside w / h, accepted?)
CONFIG_ACCEPTANCE = [
    (dict(w=128, h=64), True),
    (dict(w=128, h=64, preset=0), True), (dict(w=128, h=64, preset=1), True), (dict(w=128, h=64, preset=2), True), (dict(w=128, h=64, preset=3), True),
    (dict(w=128, h=64, preset=4), False), (dict(w=128, h=64, preset=-1), False),
    (dict(w=130, h=64), False), (dict(w=128, h=60), False), (dict(w=0, h=64), False), (dict(w=8200, h=64), False),
    (dict(w=128, h=64, bframes=2), False), (dict(w=128, h=64, bframes=7), True),
    (dict(w=128, h=64, qp=52), False), (dict(w=128, h=64, threads=9), False), (dict(w=64, h=128, threads=2), False), (dict(w=64, h=128, threads=1), True),
    (dict(w=128, h=64, input_depth=12), False), (dict(w=128, h=64, input_depth=10), True),
    (dict(w=128, h=64, inter_slice_type=1), True), (dict(w=128, h=64, inter_slice_type=2), False),
    (dict(w=128, h=64, qp_cb_offset=3), True), (dict(w=128, h=64, qp_cr_offset=-2), True), (dict(w=128, h=64, inter_slice_type=1, qp_cb_offset=-12, qp_cr_offset=12), True),
    (dict(w=128, h=64, qp_cb_offset=13), False), (dict(w=128, h=64, qp_cr_offset=-13), False),
    (dict(w=128, h=64, preset=2, qp_cb_offset=2), False), (dict(w=128, h=64, preset=3, qp_cr_offset=-1), False),
    (dict(w=128, h=64, ref=5), False), (dict(w=128, h=64, ref=2), True),
    (dict(w=128, h=64, bframes=0, keyint=0, closed_gop=True), False),
]


_PIN_ENV = {"--inter-slice-type": "XEVE_PIN_INTER_SLICE_TYPE", "--qp-cb-offset": "XEVE_PIN_QP_CB_OFFSET", "--qp-cr-offset": "XEVE_PIN_QP_CR_OFFSET"}


def app_args_and_env(cli):
    """the options the application can parse, and the environment that carries the others into the library (oracle/_ref/libxeve_param_pin.so)"""
    args, env, i = [], {}, 0
    while i < len(cli):
        if cli[i] in _PIN_ENV:
            env[_PIN_ENV[cli[i]]] = str(cli[i + 1])
            i += 2
        else:
            args.append(cli[i])
            i += 1
    if env:
        env["LD_PRELOAD"] = os.path.join(ORACLE_DIR, "_ref", "libxeve_param_pin.so")
    return args, env


def widen10(data8):
    """8-bit samples -> 10-bit samples in 16-bit little-endian words: the byte in the upper eight bits, two position-dependent bits below (so that the low bits matter)"""
    import numpy as np

    b = np.frombuffer(data8, dtype=np.uint8).astype(np.uint16)
    return ((b << 2) | ((b * 3 + np.arange(b.size, dtype=np.uint16)) & 3)).astype("<u2").tobytes()


def golden():
    return json.load(open(GOLDEN))
