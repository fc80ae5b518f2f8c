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
        subprocess.run(["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-o", OUT, SRC], check=True)
    return OUT


def _run(code_or_args, lib, timeout):
    env = dict(os.environ, LD_PRELOAD=TSAN, TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0", XW_WALK_HOST_LIB=lib, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable] + code_or_args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_the_detector_is_alive(tsan_lib):
    p = _run(["-c", "import ctypes, os; print(ctypes.CDLL(os.environ['XW_WALK_HOST_LIB']).xw_host_race_selftest())"], tsan_lib, 120)
    assert p.returncode == 0 and "WARNING: ThreadSanitizer: data race" in p.stderr, (p.stdout[-500:], p.stderr[-1500:])


def test_no_stage_of_the_walk_races(tsan_lib):
    """an I picture (two chains per team, count-only states) and a B picture with teams of 64 and 256 real threads: every result still the oracle's, and no report"""
    p = _run(["-m", "pytest", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_walk_host.py"), "-k", "real_threads"], tsan_lib, 1500)
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    assert "ThreadSanitizer" not in p.stderr, p.stderr[:6000]
