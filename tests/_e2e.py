"""End-to-end helpers: seeded synthetic YUV (SURVEY.md 8d recipe) and runs of the reference app built in oracle/_ref."""
import hashlib
import os
import random
import subprocess

from _libs import REF_APP, ROOT

SHIM = os.path.join(ROOT, "oracle", "_ref", "libxeve_hip_shim.so")  # the product's binding (shim/xeve_hip_shim.c)
SHADOW = os.path.join(ROOT, "oracle", "_ref", "libxeve_ref_shadow.so")  # the checker's interposer (oracle/ref_shadow.c: shadow mode, oracle-as-engine, per-CTU check); embeds the shim
HIP_LIB = os.path.join(ROOT, "xeve_amd", "lib", "libxeve_hip.so")

# name -> (width, height, frames, seed, extra CLI)   -- all with -m 1 (the bitstream depends on --threads)
CASES = {
    "cfg1_cif_allintra_fast": (352, 288, 8, 1234, ["--preset", "fast", "-I", "1", "-b", "0"]),
    "tiny_ldb_fast": (128, 128, 2, 7, ["--preset", "fast", "-I", "0", "-b", "0"]),
    "tiny_ra_medium": (128, 64, 4, 9, ["--preset", "medium", "-b", "1"]),
    "tiny_closed_gop": (128, 64, 8, 10, ["--preset", "fast", "--closed-gop", "-I", "4", "-b", "1"]),
    # two CTU-row worker threads (the reference calls the tables concurrently, xeve_enc.c:336-365); `-m` given last wins
    "tiny_ldb_fast_2threads": (128, 128, 2, 7, ["--preset", "fast", "-I", "0", "-b", "0", "-m", "2"]),
    # drifting texture (seed >= 5000): skip / direct / bi-prediction win here, unlike on noise
    "moving_ra_medium": (128, 64, 5, 5001, ["--preset", "medium", "-b", "1"]),
    "moving_ldb_fast": (128, 128, 3, 5002, ["--preset", "fast", "-I", "0", "-b", "0"]),
    "moving_ldb_ref3": (128, 128, 5, 5003, ["--preset", "fast", "-I", "0", "-b", "0", "--ref", "3"]),  # several reference pictures per list
    "moving_ra_b3_medium": (128, 64, 9, 5004, ["--preset", "medium", "-b", "3"]),  # hierarchical B pictures
    "jumpy_ldb_fast": (192, 128, 4, 6001, ["--preset", "fast", "-I", "0", "-b", "0"]),  # 23 x 17 samples of motion per frame
    "moving_cif_ra_medium": (352, 288, 5, 5006, ["--preset", "medium", "-b", "1"]),  # 5.5 x 4.5 CTUs: partial CTUs at the right and bottom edge
    # all-intra on structured content (the predictors matter) and on noise, two frames each: every CU of every level 64 .. 4 goes through the intra analysis
    "moving_cif_allintra_fast": (352, 288, 2, 5007, ["--preset", "fast", "-I", "1", "-b", "0"]),
    "noise_allintra_medium": (128, 128, 2, 11, ["--preset", "medium", "-I", "1", "-b", "0"]),
}

# --preset slow (xeve_enc.c:2473-2489): the quarter-pel stage of the motion search, ME range 128 and rdo_dbk_switch = 1 -- every candidate's distortion includes what
# the loop filter will do to the CU's left / top boundary (calc_delta_dist_filter_boundary).  Goldens: make_e2e_golden.py, from the unmodified reference.
SLOW_CASES = {
    "slow_tiny_ldb": (128, 128, 3, 7, ["--preset", "slow", "-I", "0", "-b", "0"]),
    "slow_moving_ra_b3": (128, 64, 9, 5004, ["--preset", "slow", "-b", "3"]),
    "slow_moving_ldb_2threads": (128, 128, 3, 5002, ["--preset", "slow", "-I", "0", "-b", "0", "-m", "2"]),
    "slow_cif_closed_gop": (352, 288, 4, 5006, ["--preset", "slow", "--closed-gop", "-I", "4", "-b", "3"]),  # partial CTUs at the right and bottom edge
    "slow_noise_allintra": (128, 128, 2, 11, ["--preset", "slow", "-I", "1", "-b", "0"]),
    "slow_jumpy_ldb": (192, 128, 4, 6001, ["--preset", "slow", "-I", "0", "-b", "0"]),
}

# --preset placebo (xeve_enc.c:2490-2506): on top of preset slow, 4x4 CUs in inter slices (inter AND intra analysis of every 4x4 node), 64x64 intra CUs in I slices, two
# reference pictures per list, the raster search behind the first diamond (me_algo 2), ME range 384, eight half- and quarter-pel positions, four merge candidates.
PLACEBO_CASES = {
    "placebo_tiny_ldb": (128, 128, 3, 7, ["--preset", "placebo", "-I", "0", "-b", "0"]),  # noise: 4x4 inter CUs are chosen
    "placebo_moving_ra_b3": (128, 64, 9, 5004, ["--preset", "placebo", "-b", "3"]),  # hierarchical B pictures, the range scaled by the POC distance
    "placebo_moving_ldb_2threads": (128, 128, 3, 5002, ["--preset", "placebo", "-I", "0", "-b", "0", "-m", "2"]),
    "placebo_cif_closed_gop": (352, 288, 4, 5006, ["--preset", "placebo", "--closed-gop", "-I", "4", "-b", "3"]),  # partial CTUs at the right and bottom edge
    "placebo_noise_allintra": (128, 128, 2, 11, ["--preset", "placebo", "-I", "1", "-b", "0"]),  # 64x64 intra CUs in I slices
    "placebo_jumpy_ldb": (192, 128, 4, 6001, ["--preset", "placebo", "-I", "0", "-b", "0"]),  # 23 x 17 samples of motion per frame: the raster search runs
    "placebo_one_ctu_ldb": (64, 64, 2, 5041, ["--preset", "placebo", "-I", "0", "-b", "0"]),  # one CTU, I + B: what the race detector's long form walks with 64 real threads
}

# BASELINE.json's configs 2, 3 and 4 at their REAL picture sizes (first frames only): goldens are made by tests/golden/make_e2e_golden.py from the reference app; the
# GPU suite encodes them once with the whole inter analysis served by the GPU (tests/test_e2e_real_sizes.py).  Not part of CASES: the CPU suite does not re-encode them.
REAL_CASES = {
    "cfg2_720p_ldb_fast": (1280, 720, 2, 2, ["--preset", "fast", "-b", "0", "-I", "0"]),  # 20 x 12 CTUs, the last row 16 samples high; low-delay B, ME range 64
    "cfg3_1080p_ra_medium": (1920, 1080, 3, 3, ["--preset", "medium"]),  # 30 x 17 CTUs (last row 56 high); default random-access GOP (-b 15): I, then B pictures by POC distance
    "cfg4_2160p_closedgop_medium": (3840, 2160, 2, 4, ["--preset", "medium", "--closed-gop", "-I", "8"]),  # 60 x 34 CTUs (last row 48 high): IDR + one inter picture
    # the same two with the reference's maximum of 8 CTU-row threads (`-m` given last wins): a different but equally deterministic bitstream (checked: two runs,
    # same md5), and eight encoder threads calling the GPU side by side -- what makes 86 000 / 173 000 per-CU calls fit the GPU suite's time budget
    "cfg3_1080p_ra_medium_m8": (1920, 1080, 3, 3, ["--preset", "medium", "-m", "8"]),
    "cfg4_2160p_closedgop_medium_m8": (3840, 2160, 2, 4, ["--preset", "medium", "--closed-gop", "-I", "8", "-m", "8"]),
}


def make_yuv(path, w, h, frames, seed):
    """seeds below 5000: i.i.d. noise (SURVEY.md 8d recipe); from 5000: a smooth texture drifting over the frames plus light noise -- content on
    which skip / merge, temporal direct and bi-prediction actually win"""
    import numpy as np

    random.seed(seed)
    if seed < 5000:  # bytes(random.getrandbits(8) ...) without the Python loop: getrandbits(8) is the top byte of one MT19937 output, and numpy's legacy generator runs
        # the same twister from the same state (tests/test_bench_inputs.py holds the two against each other)
        st = random.getstate()
        rs = np.random.RandomState()
        rs.set_state(("MT19937", np.array(st[1][:624], dtype=np.uint32), st[1][624]))
        (rs.randint(0, 2 ** 32, size=w * h * 3 // 2 * frames, dtype=np.uint32) >> 24).astype(np.uint8).tofile(path)
        return

    r = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for t in range(frames):
            for (pw, ph, sc, base) in ((w, h, 1.0, 128.0), (w // 2, h // 2, 2.0, 110.0), (w // 2, h // 2, 2.0, 140.0)):
                yy, xx = np.mgrid[0:ph, 0:pw].astype(np.float64) * sc
                xs, ys = (xx + 3.0 * t, yy + 2.0 * t) if seed < 6000 else (xx + 23.0 * t, yy + 17.0 * t)  # (seeds from 6000: jumps the first search only finds on its far rings)
                a = base + 60 * np.sin(xs / 9.0) * np.cos(ys / 7.0) + 30 * np.sin((xs + ys) / 13.0) + 12 * np.sin(xs / 3.0)
                a[ph // 2:, :] += 25 * np.sin((xx[ph // 2:, :] - 5.0 * t) / 6.0)  # a second motion in the lower half
                f.write(np.clip(a + r.integers(-2, 3, size=a.shape), 0, 255).astype(np.uint8).tobytes())


# Main-profile clips (oracle/_ref/xevem_app = the same app source on the Main library): every Main tool at its default (affine, DMVR, MMVD, ADMVP, IQT, ATS, ...);
# goldens in tests/golden/e2e_main_v1.json (tests/golden/make_e2e_main_golden.py)
MAIN_APP = os.path.join(ROOT, "oracle", "_ref", "xevem_app")
SHIM_MAIN = os.path.join(ROOT, "oracle", "_ref", "libxeve_hip_shim_main.so")
MAIN_CASES = {
    # hierarchical B pictures: bi-predicted merge candidates with references at equal distance on both sides -> DMVR (5 600 luma + 11 200 chroma calls), the
    # MMVD search's bilinear predictions (110 000), tool_iqt's 16-bit transforms (116 000) -- counted with an interposer on the reference's own tables
    "main_moving_ra_b3_fast": (64, 64, 5, 5004, ["--profile", "main", "--preset", "fast", "-b", "3"]),
    "main_moving_ldb_medium": (64, 64, 3, 5002, ["--profile", "main", "--preset", "medium", "-I", "0", "-b", "0"]),
}


# clips on which the Main encoder's adaptive loop filter ends up ON (tool_alf is the Main default, but on the tiny clips above every picture decides against it): counted
# with oracle/ref_shim_alf.c in its count-only mode -- 64 classification calls, 4 x 7x7, 8 x 5x5 on the first; 80 / 8 / 0 on the second
MAIN_ALF_CASES = {
    "main_alf_moving_q22": (128, 128, 3, 5002, ["--profile", "main", "--preset", "fast", "-I", "0", "-b", "0", "-q", "22"]),
    "main_alf_noise_q37": (128, 128, 3, 11, ["--profile", "main", "--preset", "fast", "-I", "0", "-b", "0", "-q", "37"]),
}
SHIM_ALF = os.path.join(ROOT, "oracle", "_ref", "libxeve_hip_shim_alf.so")
SHIM_AFFINE = os.path.join(ROOT, "oracle", "_ref", "libxeve_hip_shim_affine.so")  # xeve_affine_mc, called by name, forwarded to xeve_hip_affine_mc_host (oracle/ref_shim_affine.c)


def run_app_main(yuv, out, w, h, frames, extra, hip=False, timeout=1500, shim=None, env_extra=None):
    cmd = [MAIN_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-m", "1", "-v", "0", "-o", out] + list(extra)
    env = dict(os.environ)
    if env_extra:
        env.update(env_extra)
    if shim and not hip:
        env["LD_PRELOAD"] = shim
    if hip:
        env["LD_PRELOAD"], env["XEVE_HIP_LIB"] = shim or SHIM_MAIN, HIP_LIB
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    data = open(out, "rb").read()
    return hashlib.md5(data).hexdigest(), len(data), p.stderr


def run_app(yuv, out, w, h, frames, extra, hip=False, seek=None, timeout=1500, df=False, me=False, tq=False, eco=False, mc=False, inter=False, shim_env=None, resident=False, tables=True,
            intra=False, tree=False):
    cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-m", "1", "-v", "0", "-o", out] + list(extra)
    if seek is not None:
        cmd += ["--seek", str(seek)]
    env = dict(os.environ)
    checker = bool(shim_env) and any(k in shim_env for k in ("XEVE_SHIM_SHADOW_TREE", "XEVE_SHIM_TREE_ORACLE", "XEVE_SHIM_TREE_CHECK"))  # switches only the checker's interposer knows
    if shim_env:  # e.g. settings of the motion search the app has no option for (shim/xeve_hip_shim.c), for plain and GPU runs alike
        env["LD_PRELOAD"] = SHADOW if checker else SHIM
        env.update(shim_env)
    if hip:
        env["LD_PRELOAD"] = SHADOW if checker else SHIM
        env["XEVE_HIP_LIB"] = HIP_LIB
        if not tables:
            env["XEVE_HIP_SHIM_TABLES"] = "0"  # the per-call dispatch tables stay the reference's; only the coarse routes below go to the GPU
        if inter:
            env["XEVE_HIP_SHIM_INTER"] = "1"  # the whole inter analysis of a CU (ctx->fn_pinter_analyze_cu)
            if resident:
                env["XEVE_HIP_SHIM_RESIDENT"] = "1"  # planes uploaded once per picture (xeve_hip_picture_begin from ctx->fn_mode_analyze_frame)
        if intra:
            env["XEVE_HIP_SHIM_INTRA"] = "1"  # the intra analysis of a CU (ctx->fn_pintra_analyze_cu)
        if tree:
            env["XEVE_HIP_SHIM_TREE"] = str(int(tree))  # the whole mode decision of a CTU (ctx->fn_mode_analyze_lcu), one exchange per CTU: 1 = I pictures, 2 = P and B too
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
