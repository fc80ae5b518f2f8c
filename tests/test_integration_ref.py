"""End-to-end parity: the UNMODIFIED reference encoder (oracle/_ref/xeveb_app + libxeveb_ref.so, compiled in place from
/root/reference) must produce byte-identical bitstreams
  (cpu)  to the committed goldens                                     -- pins the reference build itself;
  (cpu)  when closed-GOP shards are encoded separately and concatenated -- the multi-GPU path, SURVEY.md 4(2);
  (gpu)  when every hot-path dispatch table is replaced by libxeve_hip.so's tables (LD_PRELOAD interposer
         shim/xeve_hip_shim.c = the integration INTEGRATION.md describes), with xeve_pinter.c / xeve_mode.c unchanged.
"""
import json
import os
import re

import pytest

from _e2e import CASES, REAL_CASES, SHIM, make_yuv, run_app
from _libs import REF_APP

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_v1.json")))
needs_ref = pytest.mark.skipif(not (os.path.exists(REF_APP) and os.path.exists(SHIM)), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_app_reproduces_golden_bitstreams(tmp_path, name):
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, _ = run_app(yuv, str(tmp_path / "o.evc"), w, h, n, extra)
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
def test_closed_gop_shards_concatenate_byte_identically(tmp_path):
    """xeve_amd.gop plans the shards; each is encoded by its own process; concatenation == monolithic encode."""
    from xeve_amd import gop

    w, h, n, seed, extra = CASES["tiny_closed_gop"]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    whole = str(tmp_path / "whole.evc")
    run_app(yuv, whole, w, h, n, extra)
    parts = b""
    per_rank = [gop.shards_for_rank(n, 4, r, 2) for r in range(2)]
    for sh in gop.concat_order(per_rank):
        out = str(tmp_path / ("s%d.evc" % sh.gop))
        run_app(yuv, out, w, h, sh.frames, extra, seek=sh.seek)
        parts += open(out, "rb").read()
    assert parts == open(whole, "rb").read()


@needs_ref
def test_shard_driver_concatenates_byte_identically(tmp_path):
    """xeve_amd.gop.run_shards: one encoder process per "device" slot (here plain reference processes, two side by side), shard bitstreams concatenated by
    the driver == the monolithic encode == the committed golden"""
    from xeve_amd import gop

    w, h, n, seed, extra = CASES["tiny_closed_gop"]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "-m", "1", "-v", "0"] + list(extra)
    out = str(tmp_path / "sharded.evc")
    r = gop.run_shards(cmd, yuv, out, n, 4, devices=[0, 1])
    import hashlib
    data = open(out, "rb").read()
    assert r["bytes"] == len(data) and [g for g, _, _ in r["shards"]] == [0, 1] and {d for _, d, _ in r["shards"]} == {0, 1}
    assert (hashlib.md5(data).hexdigest(), len(data)) == (GOLD["tiny_closed_gop"]["md5"], GOLD["tiny_closed_gop"]["bytes"])


@needs_ref
@pytest.mark.gpu
def test_shard_driver_with_the_gpu_routes_on(tmp_path):
    """the same driver with the encoder processes bound to the GPU (HIP_VISIBLE_DEVICES per process, whole inter analysis + resident pictures through the
    shim): two shards on the one GPU of this box, byte-identical to the monolithic golden"""
    import hashlib

    from _e2e import HIP_LIB
    from xeve_amd import gop

    w, h, n, seed, extra = CASES["tiny_closed_gop"]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "-m", "1", "-v", "0"] + list(extra)
    env = {"LD_PRELOAD": SHIM, "XEVE_HIP_LIB": HIP_LIB, "XEVE_HIP_SHIM_INTER": "1", "XEVE_HIP_SHIM_RESIDENT": "1"}
    out = str(tmp_path / "sharded.evc")
    r = gop.run_shards(cmd, yuv, out, n, 4, devices=[0], per_device=2, env=env)
    data = open(out, "rb").read()
    assert len(r["shards"]) == 2
    assert (hashlib.md5(data).hexdigest(), len(data)) == (GOLD["tiny_closed_gop"]["md5"], GOLD["tiny_closed_gop"]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium", "tiny_ldb_fast_2threads"])
def test_bitstream_identical_with_hip_tables_installed(tmp_path, name):
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000)
    assert "HIP dispatch tables installed" in err
    served = int(re.search(r"calls served by HIP: (\d+)", err).group(1))
    assert served > 10000, err  # the encode really went through the HIP tables
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with HIP tables installed"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium", "tiny_closed_gop"])
def test_bitstream_identical_with_deblocking_and_padding_on_the_gpu(tmp_path, name):
    """on top of the dispatch tables: ctx->fn_loop_filter -> xeve_hip_deblock_host, ctx->fn_picbuf_expand -> xeve_hip_picbuf_expand_host.
    Every reconstructed picture is filtered and padded by the GPU before it becomes a reference, so any deviation changes the stream."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, df=True)
    assert "loop filter and picture padding routed to the GPU" in err
    m = re.search(r"pictures deblocked on the GPU: (\d+), padded on the GPU: (\d+)", err)
    assert m and int(m.group(1)) >= n and int(m.group(2)) >= n, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the loop filter on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium"])
def test_bitstream_identical_with_motion_search_on_the_gpu(tmp_path, name):
    """on top of everything above: pi->fn_me (pinter_me_epzs) -> xeve_hip_me_epzs_host.  The vectors, costs and the mot_bits side effect
    the GPU search returns steer every inter mode decision of the encoder, so any deviation changes the stream."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, df=True, me=True)
    assert "motion search routed to the GPU" in err
    m = re.search(r"motion searches \(pinter_me_epzs\) served by the GPU: (\d+)", err)
    assert m and int(m.group(1)) > 100, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the motion search on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium", "cfg1_cif_allintra_fast"])
def test_bitstream_identical_with_rdoq_on_the_gpu(tmp_path, name):
    """ctx->fn_tq / ctx->fn_itdp: every transform block of the encode (inter and intra, luma and chroma, 2x2 ... 64x64) is transformed and
    quantised -- zero pre-test + RDOQ with the estimates the reference derived from its live CABAC state -- and reconstructed by the GPU."""
    w, h, n, seed, extra = CASES[name]
    if name.startswith("cfg1"):
        n = 1  # one intra picture is plenty (each block is staged separately)
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    ref_md5, ref_size, _ = run_app(yuv, str(tmp_path / "ref.evc"), w, h, n, extra)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, tq=True)
    m = re.search(r"quantised \(RDOQ\) on the GPU: (\d+), dequantised \+ inverse transformed: (\d+)", err)
    assert m and int(m.group(1)) > 1000 and int(m.group(2)) > 100, err
    assert (md5, size) == (ref_md5, ref_size), "bitstream differs with RDOQ on the GPU"
    if not name.startswith("cfg1"):
        assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium"])
def test_bitstream_identical_with_cabac_bit_counting_on_the_gpu(tmp_path, name):
    """ctx->fn_eco_coef in bit-count mode: the rate of every candidate the RDO weighs (inter and intra CUs, every cbf test of pinter_residue_rdo)
    comes from the GPU's arithmetic coder, continuing the encoder's live XEVE_SBAC and handing it back field for field."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, eco=True)
    m = re.search(r"coefficient bits were counted on the GPU: (\d+)", err)
    assert m and int(m.group(1)) > 1000, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the CABAC bit counting on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ra_medium", "tiny_ldb_fast_2threads"])
def test_bitstream_identical_with_everything_on_the_gpu(tmp_path, name):
    """all routes at once: dispatch tables + recon, motion search, CU prediction, transform / RDOQ / inverse, CABAC bit counting, loop filter, padding
    (the two-thread clip drives every host-memory entry point from two CTU-row workers concurrently)"""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, df=True, me=True, tq=True, eco=True, mc=True)
    for needle in ("HIP dispatch tables installed", "motion search routed", "CU motion compensation routed", "transform + RDOQ", "CABAC bit counting", "loop filter and picture padding"):
        assert needle in err, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_ldb_fast", "tiny_ra_medium"])
def test_bitstream_identical_with_cu_prediction_on_the_gpu(tmp_path, name):
    """pi->fn_mc (pinter_mc -> xeve_mc): clip, per-list interpolation of Y / U / V, identical-motion shortcut and bi-prediction average as ONE call"""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, mc=True)
    m = re.search(r"CU predictions \(xeve_mc\) made on the GPU: (\d+)", err)
    assert m and int(m.group(1)) > 500, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with xeve_mc on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moving_ra_medium", "moving_ldb_fast", "tiny_ra_medium", "tiny_ldb_fast_2threads", "moving_ldb_ref3", "moving_ra_b3_medium", "moving_cif_ra_medium", "jumpy_ldb_fast"])
def test_bitstream_identical_with_the_whole_inter_analysis_on_the_gpu(tmp_path, name):
    """ctx->fn_pinter_analyze_cu -> xeve_hip_pinter_analyze_cu_host: skip / merge analysis, temporal direct, both lists' motion searches over every
    reference picture, check_best_mvp, the iterated bi-prediction search, every pinter_residue_rdo, the mode decision and the reconstruction of
    every inter CU come from the GPU; the reference keeps the quad-tree recursion, the intra modes and the bitstream writer."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, inter=True)
    assert "whole inter analysis of a CU routed to the GPU" in err
    m = re.search(r"whole inter analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    assert m and int(m.group(1)) > 200 and int(m.group(2)) == 0, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the inter analysis on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moving_ra_medium", "moving_ldb_ref3"])
def test_bitstream_identical_with_inter_analysis_and_every_other_route_on_the_gpu(tmp_path, name):
    """the inter analysis on the GPU as one call per CU, and everything the encoder still does itself around it (intra CUs' transform / RDOQ / bit
    counting, reconstruction, loop filter, padding) through the other GPU routes"""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, inter=True, df=True, tq=True, eco=True, mc=True, me=True)
    m = re.search(r"whole inter analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    assert m and int(m.group(1)) > 200 and int(m.group(2)) == 0, err
    assert "loop filter and picture padding" in err and "transform + RDOQ" in err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name,me_env", [("jumpy_ldb_fast", {"XEVE_SHIM_ME_COMPLEXITY": "2"}), ("jumpy_ldb_fast", {"XEVE_SHIM_ME_COMPLEXITY": "2", "XEVE_SHIM_ME_LEVEL": "1"}),
                                         ("moving_ldb_ref3", {"XEVE_SHIM_ME_COMPLEXITY": "2", "XEVE_SHIM_ME_LEVEL": "1"})])
def test_bitstream_identical_with_raster_search_and_integer_refinement_on_the_gpu(tmp_path, name, me_env):
    """the branches of pinter_me_epzs no preset below placebo takes -- me_raster (me_algo 2) and me_ipel_refinement (me_sub 1) -- switched on in the
    unmodified encoder through the shim's overrides: the plain run and the runs with pi->fn_me / ctx->fn_pinter_analyze_cu on the GPU agree byte for byte
    (and the override does change the stream, so the branches are really taken)"""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    ref = run_app(yuv, str(tmp_path / "ref.evc"), w, h, n, extra, shim_env=me_env)
    assert ref[:2] != (GOLD[name]["md5"], GOLD[name]["bytes"]), "the override did not change the encode"
    got_me = run_app(yuv, str(tmp_path / "me.evc"), w, h, n, extra, hip=True, timeout=3000, me=True, shim_env=me_env)
    assert got_me[:2] == ref[:2], "bitstream differs with the raster / integer-refinement search on the GPU"
    got_inter = run_app(yuv, str(tmp_path / "inter.evc"), w, h, n, extra, hip=True, timeout=3000, inter=True, shim_env=me_env)
    m = re.search(r"whole inter analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", got_inter[2])
    assert m and int(m.group(1)) > 100 and int(m.group(2)) == 0, got_inter[2]
    assert got_inter[:2] == ref[:2], "bitstream differs with the inter analysis (raster / integer refinement inside) on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["noise_allintra_medium", "moving_cif_allintra_fast", "tiny_ldb_fast", "moving_ra_medium"])
def test_bitstream_identical_with_the_intra_analysis_on_the_gpu(tmp_path, name):
    """ctx->fn_pintra_analyze_cu -> xeve_hip_pintra_analyze_cu_host: neighbours from the picture being reconstructed, the five predictors, the candidate list, the
    luma / chroma RDO and the CU's cost + exit coder state of EVERY intra-analysed CU (all of them in the I pictures, the coded ones in P / B pictures) come
    from the GPU; each result feeds the next CU's neighbours, so any deviation changes the stream."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, intra=True, tables=False)
    assert "intra analysis of a CU routed to the GPU" in err
    m = re.search(r"CUs whose intra analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    assert m and int(m.group(1)) > 300 and int(m.group(2)) == 0, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"]), "bitstream differs with the intra analysis on the GPU"


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moving_ra_medium", "tiny_ldb_fast_2threads", "moving_cif_ra_medium"])
def test_bitstream_identical_with_inter_and_intra_analysis_on_the_gpu(tmp_path, name):
    """both analyses of every CU on the GPU: what stays with the reference is the quad-tree recursion with its cost comparisons, the bitstream writer, the
    loop filter's caller and rate control"""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "hip.evc"), w, h, n, extra, hip=True, timeout=3000, inter=True, intra=True, resident=True, df=True, tables=False)
    mi = re.search(r"CUs whose intra analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    me = re.search(r"CUs whose whole inter analysis ran on the GPU: (\d+) \(left to the reference: (\d+)\)", err)
    assert mi and me and int(mi.group(1)) > 300 and int(me.group(1)) > 300 and int(mi.group(2)) == 0, err
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])


@needs_ref
@pytest.mark.parametrize("name,nctu,ninter", [("noise_allintra_medium", 8, 0), ("moving_cif_allintra_fast", 60, 0), ("tiny_ra_medium", 8, 6), ("moving_cif_ra_medium", 150, 120),
                                             ("moving_ldb_ref3", 20, 16), ("moving_ra_b3_medium", 18, 16), ("jumpy_ldb_fast", 24, 18), ("cfg2_720p_ldb_fast", 480, 240)])
def test_oracle_ctu_mode_decision_matches_the_live_encoder(tmp_path, name, nctu, ninter):
    """xo_mode_analyze_ctu (mode_analyze_lcu -> mode_coding_tree -> mode_coding_unit, xeve_mode.c:1169-1350, 2007-2610, restated in oracle/: I, P and B slices) runs
    BESIDE the unmodified reference inside the live encoder (oracle/ref_shadow.c: ctx->fn_mode_analyze_lcu hooked in shadow mode) from the same entry state, and every
    product of the walk -- split flags, CU modes, intra modes, motion data, depths, levels, reconstruction, the context maps (units, intra modes, vectors,
    reference indices), the picture and the coder state handed on -- is compared per CTU.  CPU only: this pins the oracle the device-side tree walk is checked
    against; the CIF clips have partial CTUs at the right and bottom edges, the B clips temporal direct and bi-prediction."""
    from _libs import ORACLE_SO

    w, h, n, seed, extra = (CASES[name] if name in CASES else REAL_CASES[name])  # (the 1280x720 clip: a real picture size, the bottom CTU row cut at 720 = 11 * 64 + 16)
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    dump = str(tmp_path / "slice_data.bin")
    md5, size, err = run_app(yuv, str(tmp_path / "o.evc"), w, h, n, extra, shim_env={"XEVE_SHIM_SHADOW_TREE": ORACLE_SO, "XEVE_SHIM_SHADOW_DUMP": dump})
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])  # the shadow run leaves the encode untouched
    m = re.search(r"shadow tree walk: (\d+) CTUs compared, (\d+) differ \((\d+) not covered\), (\d+) of them in P / B", err)
    assert m, err[-800:]
    assert tuple(int(m.group(k)) for k in (1, 2, 3, 4)) == (nctu, 0, 0, ninter), err[-1500:]
    # the writer's side: xo_eco_ctu (xeve_eco_tree restated) wrote every CTU but the last of each picture beside the reference -- the coder state the next CTU enters
    # with, the bytes in the bitstream buffer, the unit flags xeve_eco_unit stores
    k = re.search(r"shadow writer: (\d+) CTUs written by the oracle beside xeve_eco_tree \((\d+) bytes of bitstream compared\), (\d+) differ", err)
    assert k and int(k.group(1)) == nctu - n and int(k.group(2)) > 0 and int(k.group(3)) == 0, err[-1500:]
    # the whole picture: at every picture's first CTU the oracle decides AND writes all CTUs on its own -- each CTU entering with the state its own writer left -- then
    # the tile's end.  Its bytes before the tile's end are what the reference's CTU loop wrote; all of them are the tail of that picture's slice NAL unit in the output
    # file: the oracle reproduces the slice data of I, P and B pictures.
    q = re.search(r"shadow pictures: (\d+) pictures decided and written by the oracle on its own \((\d+) bytes of slice data\), (\d+) differ", err)
    assert q and (int(q.group(1)), int(q.group(3))) == (n, 0), err[-1500:]
    import struct
    evc, nals, pos = open(str(tmp_path / "o.evc"), "rb").read(), [], 0
    while pos + 4 <= len(evc):  # NAL units behind a 4-byte length
        ln = struct.unpack(">I", evc[pos:pos + 4])[0]
        nals.append(evc[pos + 4:pos + 4 + ln])
        pos += 4 + ln
    assert pos == len(evc)
    d, at, used = open(dump, "rb").read(), 0, set()
    while at < len(d):
        poc, nb = struct.unpack("<ii", d[at:at + 8])
        body = d[at + 8:at + 8 + nb]
        at += 8 + nb
        hit = [i for i, x in enumerate(nals) if i not in used and x.endswith(body) and len(x) - len(body) < 64]
        assert len(hit) == 1, (name, "poc", poc, nb, hit)
        used.add(hit[0])
    assert len(used) == n


@needs_ref
@pytest.mark.parametrize("name,nctu,left", [("noise_allintra_medium", 8, 0), ("cfg1_cif_allintra_fast", 240, 0), ("tiny_ldb_fast_2threads", 8, 0), ("moving_cif_ra_medium", 150, 0),
                                            ("moving_ra_b3_medium", 18, 0), ("jumpy_ldb_fast", 24, 0)])
def test_route_adapter_of_the_ctu_mode_decision_with_the_oracle_as_engine(tmp_path, name, nctu, left):
    """the adapter that serves ctx->fn_mode_analyze_lcu from an external tree walk (shim/xeve_hip_shim.c, shim_route_mode_analyze_lcu: what it hands over and what it
    stores back into ctx->map_cu_data, the context maps and PIC_MODE) run with the ORACLE's walk as the engine: the bitstream must not change.  CPU only -- this
    tests the adapter (test infrastructure); the same adapter with the GPU as the engine is the gpu test below."""
    from _libs import ORACLE_SO

    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "o.evc"), w, h, n, extra, shim_env={"XEVE_SHIM_TREE_ORACLE": ORACLE_SO})
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])
    m = re.search(r"mode decision ran on the oracle \(CPU\): (\d+) \(left to the reference: (\d+)\)", err)
    assert m and (int(m.group(1)), int(m.group(2))) == (nctu, left), err[-800:]


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name,nctu", [("noise_allintra_medium", 8), ("moving_cif_allintra_fast", 60), ("tiny_ra_medium", 2), ("tiny_ldb_fast_2threads", 4)])
def test_bitstream_identical_with_the_ctu_mode_decision_on_the_gpu(tmp_path, name, nctu):
    """ctx->fn_mode_analyze_lcu of every I-picture CTU served by the device-side tree walk (xeve_hip_mode_analyze_ctu_intra_host): ONE exchange per CTU -- the CU
    loop, the split decisions, the map and picture updates all happen on the device -- and the bitstream is byte-identical.  P / B pictures stay with the reference
    (their CTUs are counted as left to it)."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "o.evc"), w, h, n, extra, hip=True, tables=False, tree=True)
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])
    m = re.search(r"mode decision ran on the GPU: (\d+) \(left", err)
    assert m and int(m.group(1)) == nctu, err[-800:]


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name,nctu", [("tiny_ldb_fast", 8), ("tiny_ra_medium", 8), ("moving_ra_medium", 10), ("moving_ldb_ref3", 20), ("moving_ra_b3_medium", 18), ("jumpy_ldb_fast", 24),
                                       ("tiny_ldb_fast_2threads", 8), ("moving_cif_ra_medium", 150)])
def test_bitstream_identical_with_every_ctu_decided_on_the_gpu(tmp_path, name, nctu):
    """ctx->fn_mode_analyze_lcu of EVERY CTU -- I, P and B pictures -- served by the device-side tree walk (xeve_hip_mode_analyze_ctu_host): the quad-tree, the inter and
    intra analysis of every node, the mode comparison, the map / motion / picture updates all on the device, ONE exchange per CTU; the reference keeps the frame
    loop, the entropy coder and the loop filter.  Byte-identical bitstreams; no CTU is left to the reference."""
    w, h, n, seed, extra = CASES[name]
    yuv = str(tmp_path / "in.yuv")
    make_yuv(yuv, w, h, n, seed)
    md5, size, err = run_app(yuv, str(tmp_path / "o.evc"), w, h, n, extra, hip=True, tables=False, tree=2)
    assert (md5, size) == (GOLD[name]["md5"], GOLD[name]["bytes"])
    m = re.search(r"mode decision ran on the GPU: (\d+) \(left to the reference: (\d+)\), ([0-9.]+) ms per CTU", err)
    assert m and (int(m.group(1)), int(m.group(2))) == (nctu, 0), err[-800:]
    print("%s: %d CTUs, %s ms per CTU" % (name, nctu, m.group(3)))
