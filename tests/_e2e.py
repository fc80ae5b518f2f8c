"""End-to-end helpers: seeded synthetic YUV (SURVEY.md 8d recipe) and runs of the reference app built in oracle/_ref."""
import hashlib
import os
import random
import subprocess

from _libs import REF_APP, ROOT

SHIM = os.path.join(ROOT, "oracle", "_ref", "libxeve_hip_shim.so")
HIP_LIB = os.path.join(ROOT, "xeve_amd", "lib", "libxeve_hip.so")

# name -> (width, height, frames, seed, extra CLI)   -- all with -m 1 (the bitstream depends on --threads)
CASES = {
    "cfg1_cif_allintra_fast": (352, 288, 8, 1234, ["--preset", "fast", "-I", "1", "-b", "0"]),
    "tiny_ldb_fast": (128, 128, 2, 7, ["--preset", "fast", "-I", "0", "-b", "0"]),
    "tiny_ra_medium": (128, 64, 4, 9, ["--preset", "medium", "-b", "1"]),
    "tiny_closed_gop": (128, 64, 8, 10, ["--preset", "fast", "--closed-gop", "-I", "4", "-b", "1"]),
    # two CTU-row worker threads (the reference calls the tables concurrently, xeve_enc.c:336-365); `-m` given last wins
    "tiny_ldb_fast_2threads": (128, 128, 2, 7, ["--preset", "fast", "-I", "0", "-b", "0", "-m", "2"]),
}


def make_yuv(path, w, h, frames, seed):
    random.seed(seed)
    with open(path, "wb") as f:
        f.write(bytes(random.getrandbits(8) for _ in range(w * h * 3 // 2 * frames)))


def run_app(yuv, out, w, h, frames, extra, hip=False, seek=None, timeout=1500, df=False, me=False, tq=False, eco=False, mc=False):
    cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-m", "1", "-v", "0", "-o", out] + list(extra)
    if seek is not None:
        cmd += ["--seek", str(seek)]
    env = dict(os.environ)
    if hip:
        env["LD_PRELOAD"] = SHIM
        env["XEVE_HIP_LIB"] = HIP_LIB
        if mc:
            env["XEVE_HIP_SHIM_MC"] = "1"  # also pi->fn_mc (pinter_mc -> xeve_mc), the whole CU prediction
        if eco:
            env["XEVE_HIP_SHIM_ECO"] = "1"  # also ctx->fn_eco_coef while the encoder counts bits (the RDO's rate term)
        if tq:
            env["XEVE_HIP_SHIM_TQ"] = "1"  # also ctx->fn_tq (transform + zero pre-test + RDOQ) and ctx->fn_itdp (dequant + inverse transform)
        if me:
            env["XEVE_HIP_SHIM_ME"] = "1"  # also the per-list motion search (pi->fn_me = pinter_me_epzs)
        if df:
            env["XEVE_HIP_SHIM_DF"] = "1"  # also the loop filter and the picture padding (ctx->fn_loop_filter, ctx->fn_picbuf_expand)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    data = open(out, "rb").read()
    return hashlib.md5(data).hexdigest(), len(data), p.stderr
