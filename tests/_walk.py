"""Test infrastructure: builds and loads the HOST side of xeve_amd/csrc/walk.h (the fused CTU walk that libxeve_hip.so runs as ONE kernel per CTU step, every function
__host__ __device__) as a team of one thread, so that the CPU suite compares it bit for bit with the pinned oracle.  hipcc --cuda-host-only; nothing of this is linked
into the product library."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

from _libs import ROOT, SBAC_DTYPE, c_int, c_void_p, ptr
from _tree_cases import CTU_DATA_DTYPE, CTU_JOB_DTYPE, TreeParams

SRC = os.path.join(ROOT, "tests", "native", "walk_host.cpp")
DEPS = glob.glob(os.path.join(ROOT, "xeve_amd", "csrc", "walk*.h")) + [SRC]
OUT = os.path.join(ROOT, "tests", "native", "build", "libwalk_host.so")
HIPCC = "/opt/rocm/bin/hipcc"
_lib = None


def available():
    return os.path.exists(HIPCC)


def walk():
    global _lib
    if _lib is None:
        alt = os.environ.get("XW_WALK_HOST_LIB")  # (tests/test_walk_race.py: the ThreadSanitizer build of the same harness)
        if alt:
            _lib = C.CDLL(alt)
        elif not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in DEPS):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            tmp = "%s.%d.tmp" % (OUT, os.getpid())  # (several test processes may build at once: each its own file, the rename is atomic)
            subprocess.run([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-pthread", "-o", tmp, SRC],
                           check=True)
            os.replace(tmp, OUT)
        if _lib is None:
            _lib = C.CDLL(OUT)
        _lib.xw_host_walk_mt.restype = c_int
        _lib.xw_host_walk_mt.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 9 + [c_int] + [c_void_p] * 3 + [c_int, c_int, c_int, c_int]
        _lib.xw_host_walk.restype = c_int
        _lib.xw_host_walk.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 9 + [c_int] + [c_void_p] * 3 + [c_int, c_int, c_int]
    return _lib


def host_walk(org_ptrs, s_org_l, s_org_c, mod_ptrs, s_mod_l, s_mod_c, scu, ipm, tidx, cu_mode, pic_elems, states, P, inter, jobs, chains_per_team=1, full=1, vh=0, threads=1):
    """one call of the walk over `jobs` (CTU_JOB_DTYPE records): returns (ctu data records, next_best records, costs); the planes behind mod_ptrs and the maps are updated"""
    L = walk()
    n = len(jobs)
    out, nxt, cost = np.zeros(n, CTU_DATA_DTYPE), np.zeros(n, SBAC_DTYPE), np.zeros(n, np.float64)
    org = (c_void_p * 3)(*[int(a) for a in org_ptrs])
    mod = (c_void_p * 3)(*[int(a) for a in mod_ptrs])
    pe = (C.c_int64 * 5)(*[int(v) for v in pic_elems]) if pic_elems is not None else None
    rc = L.xw_host_walk_mt(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, ptr(scu), ptr(ipm), ptr(tidx), ptr(cu_mode), pe, ptr(states), C.byref(P),
                           C.byref(inter) if inter is not None else None, ptr(jobs), n, ptr(out), ptr(nxt), ptr(cost), chains_per_team, full, vh, threads)
    assert rc == 0
    return out, nxt, cost
