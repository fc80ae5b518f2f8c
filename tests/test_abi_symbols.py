"""CPU suite: libxeve_hip.so loads and exports every symbol include/xeve_hip.h declares (no compute, no GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xeve_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    funcs = set(re.findall(r"\b(xeve_\w+)\s*\(", src))
    tables = set(re.findall(r"extern\s+const\s+\w+\s+(xeve_tbl_\w+)\s*\[", src))
    return sorted(funcs), sorted(tables)


def test_library_exports_every_declared_symbol():
    from xeve_amd import lib

    L = lib.load()
    funcs, tables = declared_symbols()
    assert len(funcs) >= 20 and len(tables) == 8
    for name in funcs + tables:
        assert C.c_void_p.in_dll(L, name) is not None, name
    # and the Python binding covers exactly the same set
    assert set(lib.FUNCTIONS) == set(funcs)
    assert set(lib.TABLES) == set(tables)


def test_tables_are_fully_populated():
    from xeve_amd import lib

    L = lib.load()
    for name, tbl in L.tables.items():
        for f in tbl:
            assert C.cast(f, C.c_void_p).value, name


def test_uninitialised_use_fails_loudly():
    """No silent fallback: the batched API refuses to run before xeve_hip_init() (and there is no GPU here)."""
    from xeve_amd import lib

    L = lib.load()
    rc = L.xeve_hip_avg(None, None, None, 0, None)
    assert rc != 0 and "xeve_hip_init" in lib.last_error()


def test_init_without_gpu_raises():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import xeve_amd

    with pytest.raises(xeve_amd.XeveHipError):
        xeve_amd.init(0)
