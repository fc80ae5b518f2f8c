"""GPU suite: BASELINE.json's configs 2, 3 and 4 at their REAL picture sizes, once through the unmodified reference encoder with the whole inter analysis
of every CU served by the GPU (ctx->fn_pinter_analyze_cu -> xeve_hip_pinter_analyze_cu_host, pictures resident in HBM: one upload per plane and picture).
Byte-identical to the goldens the reference app itself produced (tests/golden/make_e2e_golden.py).  What these sizes pin that the small clips cannot: partial
CTU rows (720 = 11*64 + 16, 1080 = 16*64 + 56, 2160 = 33*64 + 48), the MV clip window and search ranges at real picture extents, get_range_ipel's POC scaling
on the default 16-picture random-access GOP, 64x64 CUs in quantity."""
import json
import os
import re
import time

import pytest

from _e2e import CASES, REAL_CASES, SHIM, make_yuv, run_app
from _libs import REF_APP

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_v1.json")))
needs_ref = pytest.mark.skipif(not (os.path.exists(REF_APP) and os.path.exists(SHIM)), reason="oracle/_ref not built")
pytestmark = [pytest.mark.gpu, pytest.mark.gpu_last, needs_ref]


def _encode(tmp_path, name, cases, min_cus, tables=False, tree_ctus=0):
    w, h, n, seed, extra = cases[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    t0 = time.perf_counter()
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, inter=True, resident=True, tables=tables, tree=tree_ctus > 0)
    dt = time.perf_counter() - t0
    m = re.search(r"whole inter analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    r = re.search(r"resident pictures: (\d+) pictures announced, (\d+) planes uploaded \((\d+) bytes\), (\d+) plane look-ups", err)
    assert m and r, err
    cus, left = int(m.group(1)), int(m.group(2))
    t = re.search(r"time inside the GPU calls: ([0-9.]+) s = (\d+) us per CU", err)
    pics, uploads, hits = int(r.group(1)), int(r.group(2)), int(r.group(4))
    print("%s: %d CUs on the GPU in %.1f s wall (%.0f us per CU incl. the host side of the encoder; inside the GPU calls, summed over the encoder threads: %s s = %s us per CU), "
          "%d pictures, %d plane uploads, %d look-ups from HBM" % (name, cus, dt, 1e6 * dt / max(1, cus), t.group(1) if t else "?", t.group(2) if t else "?", pics, uploads, hits))
    assert cus >= min_cus and left == 0, err
    if tree_ctus:
        k = re.search(r"mode decision ran on the GPU: (\d+) \(left to the reference: (\d+)\), ([0-9.]+) ms per CTU", err)
        assert k and int(k.group(1)) == tree_ctus, err
        print("%s: %d I-picture CTUs decided on the GPU, %s ms per CTU (one exchange each)" % (name, tree_ctus, k.group(3)))
    assert pics == n and uploads <= 9 * pics and hits > 5 * cus, err  # (each inter picture: the original + at most two reference pictures, three planes each)
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the inter analysis on the GPU at %dx%d" % (w, h)


@pytest.mark.parametrize("name", ["moving_ra_medium", "tiny_ldb_fast_2threads"])
def test_small_clips_with_resident_pictures(tmp_path, name):
    """the resident-picture path on the small clips first (two encoder threads share the plane store in the second one); here with the per-call dispatch
    tables on the GPU as well, at the real sizes below they stay with the reference (the table layer is covered by its own tests)"""
    _encode(tmp_path, name, CASES, 200, tables=True)


def test_cfg2_1280x720_low_delay_fast(tmp_path):
    _encode(tmp_path, "cfg2_720p_ldb_fast", REAL_CASES, 15000)


def _encode_per_ctu(tmp_path, name, nctu):
    w, h, n, seed, extra = REAL_CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    t0 = time.perf_counter()
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, tables=False, tree=2)
    dt = time.perf_counter() - t0
    k = re.search(r"mode decision ran on the GPU: (\d+) \(left to the reference: (\d+)\), ([0-9.]+) ms per CTU", err)
    assert k and (int(k.group(1)), int(k.group(2))) == (nctu, 0), err[-800:]
    print("%s: %d CTUs decided on the GPU in %.1f s wall, %s ms per CTU (one exchange each)" % (name, nctu, dt, k.group(3)))
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@pytest.mark.gpu_full  # (46 s: the per-CTU host form at 1280x720; the default suite keeps it at 352x288, tests/test_integration_ref.py)
def test_cfg2_1280x720_with_every_ctu_decided_on_the_gpu(tmp_path):
    """the same clip with ctx->fn_mode_analyze_lcu of all 480 CTUs (the I and the P picture) served by the device-side tree walk: one exchange per CTU, nothing per CU"""
    _encode_per_ctu(tmp_path, "cfg2_720p_ldb_fast", 480)


@pytest.mark.gpu_full  # (158 s: one exchange per CTU at 1920x1080 -- parity plumbing, not a product path)
def test_cfg3_1920x1080_with_every_ctu_decided_on_the_gpu(tmp_path):
    """1920x1080 random access (I, B, B in coding order: POC-scaled search ranges, temporal direct, both ECU depths, the bottom CTU row cut at 1080 = 16 * 64 + 56):
    all 3 x 510 CTUs decided on the device"""
    _encode_per_ctu(tmp_path, "cfg3_1080p_ra_medium", 1530)


def test_cfg3_1920x1080_random_access_medium_8_threads(tmp_path):
    _encode(tmp_path, "cfg3_1080p_ra_medium_m8", REAL_CASES, 60000)


# (round 3: the one-thread run of config 3 and the 3840x2160 run through the per-CU route -- 160 s of GPU suite for a route that is parity plumbing -- made room for the
# batch encoder's runs of configs 2, 3 and 4 in tests/test_enc_gpu.py: the same pictures, every stage on the device; the per-CU route keeps its 720p and 8-thread 1080p runs)
