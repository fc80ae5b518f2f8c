"""Race detection for the fused CTU walk (xeve_amd/csrc/walk.h): its stages exchange data through the team's shared memory and the per-chain workspace with a barrier
between producer and consumer -- on the device a missing barrier is a rare wrong bit, here it is a ThreadSanitizer report.  The harness of tests/native/walk_host.cpp is built
once more with `g++ -fsanitize=thread` and the team-of-real-threads tests of tests/test_walk_host.py (sync() = a pthread barrier, 64 and 256 lanes, the device's lane mapping)
run in a process with the sanitizer's runtime preloaded.  The detector is proven alive first (two threads writing one word must be reported)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "walk_host.cpp")
OUT = os.path.join(ROOT, "tests", "native", "build", "libwalk_host_tsan.so")
DEPS = glob.glob(os.path.join(ROOT, "xeve_amd", "csrc", "walk*.h")) + [SRC]


def _tsan_runtime():
    try:
        p = subprocess.run(["g++", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None
    except Exception:
        return None


TSAN = _tsan_runtime()
pytestmark = pytest.mark.skipif(TSAN is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"), reason="no ThreadSanitizer runtime / HIP headers")


@pytest.fixture(scope="module")
def tsan_lib():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in DEPS):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = "%s.%d.tmp" % (OUT, os.getpid())
        subprocess.run(["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-o", tmp, SRC], check=True)
        os.replace(tmp, OUT)
    return OUT


def _run(code_or_args, lib, timeout):
    env = dict(os.environ, LD_PRELOAD=TSAN, TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0", XW_WALK_HOST_LIB=lib, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable] + code_or_args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_the_detector_is_alive(tsan_lib):
    p = _run(["-c", "import ctypes, os; print(ctypes.CDLL(os.environ['XW_WALK_HOST_LIB']).xw_host_race_selftest())"], tsan_lib, 120)
    assert p.returncode == 0 and "WARNING: ThreadSanitizer: data race" in p.stderr, (p.stdout[-500:], p.stderr[-1500:])


def test_no_stage_of_the_walk_races(tsan_lib):
    """an I picture (two chains per team, count-only states) and a B picture with teams of 64 and 256 real threads: every result still the oracle's, and no report"""
    p = _run(["-m", "pytest", "-q", "-s", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_walk_host.py"), "-k", "real_threads"], tsan_lib, 1500)
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    assert "ThreadSanitizer" not in p.stderr, p.stderr[:6000]


ENC_OUT = os.path.join(ROOT, "tests", "native", "build", "libxeve_enc_oracle_tsan.so")


@pytest.mark.skipif(not os.environ.get("XEVE_RACE_TESTS"), reason="XEVE_RACE_TESTS=1: a whole encode with every team as 64 real threads under ThreadSanitizer takes tens of minutes")
def test_no_stage_races_in_whole_encodes(tsan_lib):
    """the batch encoder's frame loop on the CPU engine (oracle/enc_oracle.cpp) with every CTU decided by the fused walk run by a team of 64 real threads: two and three row
    chains per team through I and B pictures of moving and noise content -- the reference's bitstreams, and no report"""
    tmp = os.path.join(os.path.dirname(ENC_OUT), "enc_oracle_c_tsan.o")
    subprocess.run(["gcc", "-fsanitize=thread", "-O1", "-g", "-std=c99", "-fPIC", "-ffp-contract=off", "-c", os.path.join(ROOT, "oracle", "xeve_oracle.c"), "-o", tmp], check=True)
    subprocess.run(["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-o", ENC_OUT, os.path.join(ROOT, "oracle", "enc_oracle.cpp"),
                    tmp, "-lm"], check=True)
    env = dict(os.environ, LD_PRELOAD=TSAN, TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0", XO_ENC_ORACLE_LIB=ENC_OUT, XO_ENC_WALK_THREADS="64", PYTHONPATH=ROOT)
    # (round 5: + one CTU of an I and a B picture at preset placebo -- walk_dbk.h's edge-position stage, the quarter-pel and raster searches, 4x4 inter CUs)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-s", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_walk_host.py"), "-k",
                        "(closed_gop_batches and (gops_128x128_moving_m2 or gops_192x256_noise_m3)) or placebo_one_ctu_ldb"], capture_output=True, text=True, timeout=6 * 3600, cwd=ROOT,
                       env=env)
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    assert "ThreadSanitizer" not in p.stderr, p.stderr[:6000]
