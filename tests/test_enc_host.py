"""The host side of the closed-GOP batch encoder (xeve_amd/csrc/enc_plan.h + enc_host.h: the frame loop, the reference-picture bookkeeping, QPs / lambdas / search
parameters per picture, the row chains, parameter sets + SEI + slice NAL units) pinned WITHOUT a GPU: the same template the library instantiates with its HIP engine
is instantiated with a CPU engine on the oracle (oracle/enc_oracle.cpp, test infrastructure) and must reproduce the reference application's bitstreams byte for byte
-- the committed md5s of tests/golden/e2e_v1.json (made by make_e2e_golden.py) and enc_v1.json (make_enc_golden.py), both recorded from the unmodified reference."""
import json
import os

import pytest

import _e2e
import _enc


@pytest.fixture(scope="module")
def yuv_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("enc_yuv")


def _frames(yuv_dir, name, w, h, n, seed):
    p = os.path.join(yuv_dir, name + ".yuv")
    if not os.path.exists(p):
        _e2e.make_yuv(p, w, h, n, seed)
    return open(p, "rb").read()


def test_frame_loop_matches_the_reference_applications_picture_table():
    """57 (options, frame count) pairs: coding order, temporal ids, slice types, QPs and the first reference of each list, as the application reports them"""
    plans = _enc.golden()["plans"]
    assert len(plans) >= 50
    for p in plans:
        mine = [[r[1], r[3], "BPI"[r[2]], r[4], r[6], r[7]] for r in _enc.plan_cpu(_enc.config(64, 64, p["cli"]), p["frames"])]
        assert mine == p["rows"], (p["cli"], p["frames"])
        frames = sorted(r[0] for r in _enc.plan_cpu(_enc.config(64, 64, p["cli"]), p["frames"]))
        assert frames == list(range(p["frames"]))  # every input frame coded exactly once


E2E = json.load(open(os.path.join(_enc.ROOT, "tests", "golden", "e2e_v1.json")))


@pytest.mark.parametrize("name", sorted(_e2e.CASES))
def test_single_runs_reproduce_the_reference_bitstreams(name, yuv_dir):
    """the 13 clips of tests/_e2e.py (all-intra, low-delay B incl. 3 reference pictures, random access with 1 and 3 B pictures, closed GOP, two row chains, CIF with
    partial CTUs): the whole file -- parameter sets, SEI text, every slice NAL unit -- is the reference's"""
    w, h, n, seed, cli = _e2e.CASES[name]
    out = _enc.encode_cpu(_enc.config(w, h, cli), [_frames(yuv_dir, name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.BATCH_CASES))
def test_batches_of_closed_gops_reproduce_the_reference_run_per_gop(name, yuv_dir):
    """G closed GOPs in lockstep, 1 .. 8 row chains per picture (the second writer pass included): every GOP's bitstream = the reference application's run over
    that GOP's frames (--seek g * F --frames F)"""
    w, h, gops, frames, seed, cli, threads = _enc.BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]
    whole = b"".join(outs)  # and the concatenation = the reference's ONE run over the whole sequence (what makes closed GOPs shardable)
    assert (len(whole), _enc.md5(whole)) == (g["whole"]["bytes"], g["whole"]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.DEPTH10_CASES))
def test_ten_bit_input_is_handed_to_the_codec_as_it_is(name, yuv_dir):
    """the application's -d 10: 16-bit samples, no conversion -- every GOP = the reference's run over it, and the concatenation = its one run over the sequence"""
    w, h, gops, frames, seed, cli, threads = _enc.DEPTH10_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _enc.widen10(_frames(yuv_dir, name, w, h, gops * frames, seed)), w * h * 3 * frames  # (two bytes per sample)
    cfg = _enc.config(w, h, cli, threads)
    assert cfg.reserved[1] == 10
    outs = _enc.encode_cpu(cfg, [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]
    whole = b"".join(outs)
    assert (len(whole), _enc.md5(whole)) == (g["whole"]["bytes"], g["whole"]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.HOST_PINNED_CASES))
def test_p_slices_and_chroma_qp_offsets_on_the_host_side(name, yuv_dir):
    """options the reference application cannot parse, set on the way into the reference LIBRARY (oracle/ref_param_pin.c): the frame loop, the reference lists of P
    pictures, the chroma QPs / lambdas / slice header offsets reproduce its bitstreams (the device: tests/test_enc_gpu.py)"""
    w, h, gops, frames, seed, cli, threads = _enc.HOST_PINNED_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]
    if "whole" in g:
        whole = b"".join(outs)
        assert (len(whole), _enc.md5(whole)) == (g["whole"]["bytes"], g["whole"]["md5"])


@pytest.mark.parametrize("name", sorted(_e2e.SLOW_CASES))
def test_preset_slow_single_runs(name, yuv_dir):
    """--preset slow: the search's quarter-pel stage, ME range 128 and rdo_dbk_switch = 1 -- every candidate's distortion includes what the loop filter will do to the CU's
    top and left edge (oracle/xeve_oracle.c xo_delta_dist = calc_delta_dist_filter_boundary) -- low delay, random access with hierarchical B pictures, closed GOPs with
    partial CTUs, all-intra, two row chains: the reference application's bitstreams"""
    w, h, n, seed, cli = _e2e.SLOW_CASES[name]
    threads = int(cli[cli.index("-m") + 1]) if "-m" in cli else 1
    cli = [a for i, a in enumerate(cli) if a != "-m" and (i == 0 or cli[i - 1] != "-m")]
    out = _enc.encode_cpu(_enc.config(w, h, cli, threads), [_frames(yuv_dir, name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.SLOW_BATCH_CASES))
def test_preset_slow_batches_of_closed_gops(name, yuv_dir):
    """... and as closed GOPs in lockstep with 3 / 8 row chains: a CU at the top of a CTU row filters against CTUs another chain's writer has been through"""
    w, h, gops, frames, seed, cli, threads = _enc.SLOW_BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


@pytest.mark.parametrize("name", sorted(_e2e.PLACEBO_CASES))
def test_preset_placebo_single_runs(name, yuv_dir):
    """--preset placebo: preset slow plus 4x4 CUs in inter slices (both analyses at every 4x4 node: 2x2 chroma blocks, the 4x4 SAD and Hadamard), 64x64 intra CUs in I
    slices, two reference pictures per list, the raster search behind the first diamond, ME range 384, eight sub-pel positions per stage, four merge candidates --
    the reference application's bitstreams"""
    w, h, n, seed, cli = _e2e.PLACEBO_CASES[name]
    threads = int(cli[cli.index("-m") + 1]) if "-m" in cli else 1
    cli = [a for i, a in enumerate(cli) if a != "-m" and (i == 0 or cli[i - 1] != "-m")]
    out = _enc.encode_cpu(_enc.config(w, h, cli, threads), [_frames(yuv_dir, name, w, h, n, seed)], n)[0]
    assert (len(out), _enc.md5(out)) == (E2E[name]["bytes"], E2E[name]["md5"])


@pytest.mark.parametrize("name", sorted(_enc.PLACEBO_BATCH_CASES))
def test_preset_placebo_batches_of_closed_gops(name, yuv_dir):
    w, h, gops, frames, seed, cli, threads = _enc.PLACEBO_BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


def test_preset_slow_with_chroma_qp_offsets_is_refused():
    c = _enc.config(128, 64, ["--preset", "slow", "--qp-cb-offset", "2"])
    with pytest.raises(RuntimeError, match="chroma qp offsets"):
        _enc.encode_cpu(c, [bytes(128 * 64 * 3 // 2)], 1)


@pytest.mark.parametrize("name", sorted(_enc.HEADER_OPTION_CASES))
def test_header_only_options(name, yuv_dir):
    """--info 0 (no SEI with the option list) and --level-idc: parameter sets and SEI only"""
    w, h, gops, frames, seed, cli, threads = _enc.HEADER_OPTION_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    outs = _enc.encode_cpu(_enc.config(w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
    assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]


def test_the_product_library_takes_p_slices_and_chroma_qp_offsets_inside_their_ranges():
    """xeve_hip_enc_footprint runs the product's own configuration check (no device needed): since round 5 the options are coded on the device too (tests/test_enc_gpu.py)"""
    from xeve_amd import encode, lib

    for kw in (dict(inter_slice_type=1), dict(qp_cb_offset=3), dict(qp_cr_offset=-2), dict(inter_slice_type=1, qp_cb_offset=-12, qp_cr_offset=12)):
        assert encode.footprint(encode.config(128, 64, keyint=8, closed_gop=True, **kw), 1, 2)[0] > 0
    for kw in (dict(inter_slice_type=2), dict(qp_cb_offset=13), dict(qp_cr_offset=-13)):
        with pytest.raises(lib.XeveHipError):
            encode.footprint(encode.config(128, 64, keyint=8, closed_gop=True, **kw), 1, 2)


@pytest.mark.parametrize("kw,accepted", _enc.CONFIG_ACCEPTANCE, ids=[",".join("%s=%s" % i for i in k.items()) for k, _ in _enc.CONFIG_ACCEPTANCE])
def test_the_product_library_accepts_and_refuses_what_the_table_says(kw, accepted):
    """The table tests/test_enc_gpu.py reads on the device, held here without one (the refusal test went stale when preset placebo was accepted and
    nothing in the build container noticed): xeve_hip_enc_footprint runs Param::finish, the same check xeve_hip_enc_create runs"""
    from xeve_amd import encode, lib

    kw = dict(kw)
    kw.setdefault("keyint", 8), kw.setdefault("closed_gop", True)
    c = encode.config(kw.pop("w"), kw.pop("h"), **kw)
    if accepted:
        assert encode.footprint(c, 1, 2)[0] > 0
    else:
        with pytest.raises(lib.XeveHipError):
            encode.footprint(c, 1, 2)


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import _e2e, _enc
from xeve_amd import gop
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
name, work = sys.argv[2], sys.argv[3]
W, H, gops, frames, seed, cli, threads = _enc.BATCH_CASES[name]
yuv = os.path.join(work, "in.yuv")
if r == 0:
    _e2e.make_yuv(yuv, W, H, gops * frames, seed)
dist.barrier()
data, fb = open(yuv, "rb").read(), W * H * 3 // 2 * frames
mine = [s.gop for s in gop.shards_for_rank(gops * frames, frames, r, w)]            # GOP g -> rank g mod world: the driver's partition
outs = _enc.encode_cpu(_enc.config(W, H, cli, threads), [data[g * fb:(g + 1) * fb] for g in mine], frames)   # this rank's batch, in lockstep
for g, o in zip(mine, outs):
    open(os.path.join(work, "gop%03d.evc" % g), "wb").write(o)
sizes = torch.zeros(gops, dtype=torch.int64)
for g, o in zip(mine, outs):
    sizes[g] = len(o)
dist.all_reduce(sizes)                                                              # control plane only: the data path has no collective
dist.barrier()
if r == 0:
    whole = b"".join(open(os.path.join(work, "gop%03d.evc" % g), "rb").read() for g in range(gops))
    gold = _enc.golden()["batches"][name]
    assert sizes.tolist() == [p["bytes"] for p in gold["per_gop"]], sizes
    assert (len(whole), _enc.md5(whole)) == (gold["whole"]["bytes"], gold["whole"]["md5"])
    print("OK", sizes.tolist())
dist.destroy_process_group()
"""


def test_two_ranks_encode_their_gops_and_the_concatenation_is_the_reference_stream(tmp_path):
    """the N > 1 path on the CPU (gloo, world size 2): every rank runs the batch encoder's frame loop over its own closed GOPs (rank = GOP mod world, as bench.py --gpus N
    and xeve_amd/gop.py shard them), nothing but sizes crosses ranks, and the concatenated bitstreams are the reference's single run over the whole sequence"""
    import socket
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          str(script), _enc.ROOT, "gops_128x64_noise", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_one_chain_through_the_second_writer_pass_gives_the_same_bytes(yuv_dir):
    """with one row chain the bytes of the first pass ARE the slice data (the shortcut the encoder takes); writing them again after the loop filter, as the reference
    does (xeve_enc.c:466-560), must not change a byte"""
    w, h, n, seed, cli = _e2e.CASES["tiny_closed_gop"]
    f = _frames(yuv_dir, "tiny_closed_gop", w, h, n, seed)
    a = _enc.encode_cpu(_enc.config(w, h, cli), [f], n)[0]
    b = _enc.encode_cpu(_enc.config(w, h, cli), [f], n, always_rewrite=True)[0]
    assert a == b and _enc.md5(a) == E2E["tiny_closed_gop"]["md5"]


@pytest.mark.parametrize("name", ["gops_128x64_noise", "gops_128x128_moving_m2"])
def test_flushing_after_every_picture_changes_no_byte_and_cuts_at_access_units(name, yuv_dir):
    """BatchEncoder::flush (the product's xeve_hip_enc_flush): the access unit of the picture just ended appended at once instead of at the next picture's end -- one and
    two row chains (the second writer pass pending at the cut).  The final streams are the reference's; GOP 0's stream after cut k is a prefix of it and ends where an
    access unit ends (the next bytes are a NAL unit's length field, and the lengths walk exactly to the cut)"""
    import struct

    w, h, gops, frames, seed, cli, threads = _enc.BATCH_CASES[name]
    g = _enc.golden()["batches"][name]
    data, fb = _frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
    chunks = [data[i * fb:(i + 1) * fb] for i in range(gops)]
    for per in (1, 3):
        outs, after = _enc.encode_cpu_flushed(_enc.config(w, h, cli, threads), chunks, frames, per)
        assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]]
        assert len(after) == -(-frames // per) and after == sorted(after) and len(set(after)) == len(after) and after[-1] == len(outs[0])
        ends, pos = set(), 0
        while pos < len(outs[0]):  # the application's output: a 4-byte big-endian length in front of every NAL unit
            pos += 4 + struct.unpack_from(">I", outs[0], pos)[0]
            ends.add(pos)
        assert pos == len(outs[0]) and set(after) <= ends


def test_configurations_outside_the_supported_set_are_refused():
    for bad in (dict(w=130), dict(preset=4), dict(preset=-1), dict(bframes=2), dict(threads=9), dict(inter_slice_type=2)):
        c = _enc.config(128, 64, ["--preset", "fast"])
        for k, v in bad.items():
            setattr(c, k, v)
        with pytest.raises(RuntimeError):
            _enc.encode_cpu(c, [bytes(128 * 64 * 3 // 2)], 1)
    # a picture one CTU wide with several row chains: the reference's row threads do not wait for each other there (xeve_enc.c:130-133) and its own output changes from run
    # to run (tests/golden/fuzz_enc_host.py found it) -- refused rather than answered
    c = _enc.config(64, 136, ["--preset", "fast"], threads=2)
    with pytest.raises(RuntimeError, match="one CTU wide"):
        _enc.encode_cpu(c, [bytes(64 * 136 * 3 // 2)], 1)
    # low-delay closed GOPs with an odd keyint over more than one GOP: the reference fetches stale input slots (xeve_enc.c:661 vs :1080) -- one GOP of it is fine
    e = _enc.config(128, 64, ["--preset", "fast", "-b", "0", "--closed-gop", "-I", "5"])
    with pytest.raises(RuntimeError, match="even keyint"):
        _enc.encode_cpu(e, [bytes(128 * 64 * 3 // 2 * 6)], 6)
    # more reference pictures than the frame loop's and the device path's tables hold (5 / 4 per list): refused before anything indexes them
    for ref in (5, 6, 15, -1):
        r = _enc.config(128, 64, ["--preset", "fast"])
        r.ref = ref
        with pytest.raises(RuntimeError, match="ref must lie"):
            _enc.encode_cpu(r, [bytes(128 * 64 * 3 // 2 * 2)], 2)
    # a low-delay closed GOP without an intra period: the reference divides by it (xeve_enc.c:1115-1117) -- refused, not a SIGFPE
    z = _enc.config(128, 64, ["--preset", "fast", "-b", "0", "--closed-gop", "-I", "0"])
    with pytest.raises(RuntimeError, match="keyint > 0"):
        _enc.encode_cpu(z, [bytes(128 * 64 * 3 // 2 * 2)], 2)
    d = _enc.config(128, 64, ["--preset", "fast", "-d", "10"])
    d.reserved[1] = 12
    with pytest.raises(RuntimeError, match="input depth"):
        _enc.encode_cpu(d, [bytes(128 * 64 * 3)], 1)
