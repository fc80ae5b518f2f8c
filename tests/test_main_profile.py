"""Main-profile first slice through the HIP library:
  (gpu) the five *_hip tables against the committed golden outputs of the reference's own Main tables and against the oracle;
  (cpu) the Main-profile reference app reproduces its golden bitstreams (pins the build);
  (gpu) the UNMODIFIED Main-profile encoder with xeve_hip_install_tables_main() applied by the LD_PRELOAD interposer oracle/ref_shim_main.c
        writes the byte-identical bitstream while DMVR, the MMVD search's bilinear predictions and tool_iqt's transforms run on the GPU."""
import json
import os
import re

import numpy as np
import pytest

from _e2e import MAIN_ALF_CASES, MAIN_APP, MAIN_CASES, SHIM_AFFINE, SHIM_ALF, SHIM_MAIN, make_yuv, run_app_main
from _main_cases import HIP_NAMES, OracleMain, TableMain, check_golden, run_all

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_main_v1.json")))
needs_ref = pytest.mark.skipif(not (os.path.exists(MAIN_APP) and os.path.exists(SHIM_MAIN)), reason="oracle/_ref (Main profile) not built")


@pytest.fixture(scope="module")
def hip_tables():
    import xeve_amd
    from xeve_amd import lib

    xeve_amd.init(0)
    return TableMain(lib.load(), HIP_NAMES)


@pytest.mark.gpu
def test_hip_main_tables_match_golden(hip_tables):
    from xeve_amd import lib

    L = lib.load()
    before = L.xeve_hip_table_calls_main()
    assert check_golden(hip_tables) > 200000
    assert L.xeve_hip_table_calls_main() - before > 500  # every case went through a Main-profile HIP entry


@pytest.mark.gpu
def test_hip_main_tables_match_oracle(hip_tables):
    a, b = run_all(OracleMain()), run_all(hip_tables)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), i


@pytest.mark.gpu
def test_hip_main_tables_touch_only_the_reference_footprint(hip_tables):
    """the staging reads exactly the samples the reference variant reads: a block at the very corner of an allocation (bilinear: one sample / row beyond the
    block and nothing before it; DMVR: 3 / 4 around) must not fault and must match the oracle"""
    O = OracleMain()
    r = np.random.default_rng(5)
    for kind, back, fwd in ((2, 0, 1), (0, 3, 4), (1, 1, 2)):
        w = h = 8
        s = w + back + fwd
        plane = r.integers(0, 1024, size=(h + back + fwd, s)).astype(np.int16)
        for fx, fy in ((1, 1), (1, 0), (0, 1), (0, 0)):
            gx, gy = (5 if fx else 0), (9 if fy else 0)
            a, b = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
            O.mc(kind, fx, fy, plane, back * s + back, gx, gy, s, w, a, w, h, 10)
            hip_tables.mc(kind, fx, fy, plane, back * s + back, gx, gy, s, w, b, w, h, 10)
            assert np.array_equal(a, b), (kind, fx, fy)


@needs_ref
@pytest.mark.parametrize("name", sorted(MAIN_CASES) + sorted(MAIN_ALF_CASES))
def test_main_reference_app_reproduces_golden_bitstreams(tmp_path, name):
    w, h, n, seed, extra = (MAIN_CASES.get(name) or MAIN_ALF_CASES[name])
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, _ = run_app_main(yuv, str(tmp_path / "o.evc"), w, h, n, extra)
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.gpu_full) if n == "main_moving_ra_b3_fast" else n for n in sorted(MAIN_CASES)])  # (36 s: the second Main clip)
def test_main_bitstream_identical_with_hip_tables_installed(tmp_path, name):
    w, h, n, seed, extra = MAIN_CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000)
    assert "Main-profile entries included (18 pointers)" in err
    m = re.search(r"calls served by HIP: (\d+), of them by the Main-profile entries: (\d+)", err)
    assert m and int(m.group(2)) > 50000 and int(m.group(1)) > int(m.group(2)), err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "Main-profile bitstream differs with the HIP tables installed"


def _alf_counts(err):
    m = re.search(r"ALF calls ([^:]+): classification (\d+), 7x7 filter (\d+), 5x5 filter (\d+), statistics (\d+)", err)
    assert m, err
    return m.group(1), tuple(int(m.group(i)) for i in (2, 3, 4, 5))


@needs_ref
@pytest.mark.skipif(not os.path.exists(SHIM_ALF), reason="oracle/_ref/libxeve_hip_shim_alf.so not built")
@pytest.mark.parametrize("name", sorted(MAIN_ALF_CASES))
def test_the_alf_clips_reach_classification_and_filters(tmp_path, name):
    """(cpu) the interposer in its count-only mode: the reference's own ALF functions, counted -- the clips the GPU test runs do filter"""
    w, h, n, seed, extra = MAIN_ALF_CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "o.evc"), w, h, n, extra, shim=SHIM_ALF, env_extra={"XEVE_HIP_SHIM_ALF_COUNT": "1"})
    how, (ncls, n7, n5, nst) = _alf_counts(err)
    assert "reference" in how and ncls >= 48 and n7 >= 4 and (n5 >= 8 or name != "main_alf_moving_q22") and nst >= 48
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SHIM_ALF), reason="oracle/_ref/libxeve_hip_shim_alf.so not built")
@pytest.mark.parametrize("name", sorted(MAIN_ALF_CASES))
def test_main_bitstream_identical_with_the_alf_kernels_on_the_gpu(tmp_path, name):
    """the UNMODIFIED Main-profile encoder with its ADAPTIVE_LOOP_FILTER object's three function pointers bound to the HIP host forms (oracle/ref_shim_alf.c, as
    INTEGRATION.md shows): classification of every tile and CTU, the correlation statistics of every CTU, component and filter shape (xeve_alf_get_blk_stats, interposed by
    name), the 7x7 luma and 5x5 chroma filters of every enabled CTU on the GPU -- the same bitstream"""
    w, h, n, seed, extra = MAIN_ALF_CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, shim=SHIM_ALF, timeout=600)
    how, (ncls, n7, n5, nst) = _alf_counts(err)
    assert how == "served by HIP" and ncls >= 48 and n7 >= 4 and nst >= 48
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "Main-profile bitstream differs with the ALF kernels on the GPU"


def _affine_counts(err):
    m = re.search(r"xeve_affine_mc calls ([^:]+): (\d+) \(of (\d+)\)", err)
    assert m, err
    return m.group(1), int(m.group(2)), int(m.group(3))


def _affine_search_counts(err):
    m = re.search(r"affine gradient searches ([^:]+): (\d+) \(of (\d+)\)", err)
    assert m, err
    return m.group(1), int(m.group(2)), int(m.group(3))


AFFINE_CLIPS = {**MAIN_CASES, **MAIN_ALF_CASES}  # (tool_affine is a Main default: every Main clip of the suite calls xeve_affine_mc a thousand times or more)


@needs_ref
@pytest.mark.skipif(not os.path.exists(SHIM_AFFINE), reason="oracle/_ref/libxeve_hip_shim_affine.so not built")
@pytest.mark.parametrize("name", sorted(AFFINE_CLIPS))
def test_the_main_clips_reach_affine_motion_compensation(tmp_path, name):
    """(cpu) the interposer in its count-only mode: the reference's own xeve_affine_mc, counted -- the clips the GPU test runs do predict affine CUs"""
    w, h, n, seed, extra = AFFINE_CLIPS[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "o.evc"), w, h, n, extra, shim=SHIM_AFFINE, env_extra={"XEVE_HIP_SHIM_AFFINE_COUNT": "1"})
    how, served, calls = _affine_counts(err)
    assert "reference" in how and calls >= 900 and served == calls
    how, served, calls = _affine_search_counts(err)  # (pi->fn_affine_me bound to a counting forwarder: the affine gradient searches of CUs of 16x16 and more)
    assert "reference" in how and calls >= 300 and served == calls
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SHIM_AFFINE), reason="oracle/_ref/libxeve_hip_shim_affine.so not built")
@pytest.mark.parametrize("name", sorted(AFFINE_CLIPS))
def test_main_bitstream_identical_with_affine_motion_compensation_on_the_gpu(tmp_path, name):
    """round 6 (first half): the UNMODIFIED Main-profile encoder with every xeve_affine_mc call -- the affine merge candidates, every round of the affine
    gradient search, the affine bi-prediction (xevem_pinter.c:1947, 4659, 4918) -- served by xeve_hip_affine_mc_host (oracle/ref_shim_affine.c interposes the symbol): the
    affine search around it stays the reference's and decides from the GPU's predictions -- the same bitstream"""
    w, h, n, seed, extra = AFFINE_CLIPS[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, shim=SHIM_AFFINE, timeout=900)
    how, served, calls = _affine_counts(err)
    assert how == "served by HIP" and served == calls and calls >= 900
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "Main-profile bitstream differs with affine motion compensation on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SHIM_AFFINE), reason="oracle/_ref/libxeve_hip_shim_affine.so not built")
@pytest.mark.parametrize("name", sorted(AFFINE_CLIPS))
def test_main_bitstream_identical_with_the_affine_gradient_search_on_the_gpu(tmp_path, name):
    """round 6 (second half): the UNMODIFIED Main-profile encoder with every thread's pi->fn_affine_me (the reference binds its static
    pinter_affine_me_gradient, xevem_pinter.c:6285) bound to xeve_hip_affine_me_host -- a whole search per call on the GPU: the start compensation, every round's normal
    equations, solve_equal and update, the SATD comparison -- and xeve_affine_mc on the GPU as well (oracle/ref_shim_affine.c): the same bitstream"""
    w, h, n, seed, extra = AFFINE_CLIPS[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, shim=SHIM_AFFINE, timeout=900, env_extra={"XEVE_HIP_SHIM_AFFINE_ME": "1"})
    assert "HIP affine gradient search bound" in err, err
    how, served, calls = _affine_search_counts(err)
    assert how == "served by HIP" and served == calls and calls >= 300
    how, served, calls = _affine_counts(err)
    assert how == "served by HIP" and served == calls and calls >= 900
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "Main-profile bitstream differs with the affine gradient search on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SHIM_AFFINE), reason="oracle/_ref/libxeve_hip_shim_affine.so not built")
@pytest.mark.parametrize("name", ["main_alf_noise_q37", "main_moving_ra_b3_fast"])
def test_every_affine_search_of_a_live_encode_matches_the_references_own(tmp_path, name):
    """the interposer's verify mode: the reference's own pinter_affine_me_gradient behind the GPU's on EVERY search of the live encoder (its real originals, bi-prediction
    targets, predictors, lambdas) -- vectors and value compared call by call; the search's route alone (xeve_affine_mc stays the reference's)"""
    w, h, n, seed, extra = AFFINE_CLIPS[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app_main(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, shim=SHIM_AFFINE, timeout=900,
                                  env_extra={"XEVE_HIP_SHIM_AFFINE_ME": "1", "XEVE_HIP_SHIM_AFFINE_VERIFY": "1", "XEVE_HIP_SHIM_AFFINE_MC_OFF": "1"})
    how, served, calls = _affine_search_counts(err)
    assert how == "served by HIP" and served == calls and calls >= 300
    assert "VERIFY:" not in err, [l for l in err.splitlines() if "VERIFY" in l]
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])
