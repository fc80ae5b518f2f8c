"""Batches of the closed-GOP encoder: what a batch costs (xeve_hip_enc_footprint -- host arithmetic, no device), how a job larger than a batch is cut
(xeve_amd/encode.py plan_batches), and -- on the GPU -- that GOPs coded in several batches side by side, in several rounds, are the reference's bitstreams."""
import pytest

import _enc
from xeve_amd import encode


def _cfg(w, h, threads=8):
    return encode.config(w, h, qp=32, keyint=8, bframes=15, closed_gop=True, preset="medium", threads=threads)


def test_a_batch_ends_where_its_halved_offsets_into_the_stacked_originals_reach_32_bits():
    """the library's own job records count PAIRS of samples (xh_common.h XH_OFF2_HALF): 2^33 samples of stacked originals per batch -- 896 pictures of 3840x2160, more
    than a GPU's HBM holds of 8-frame GOPs, so a GPU's job is ONE batch"""
    c = _cfg(3840, 2160)
    assert encode.footprint(c, 1, 2)[1] == 896  # (2496 rows x 3840 samples per stacked picture)
    assert encode.footprint(_cfg(1920, 1080), 1, 2)[1] == (2 ** 33 - 1) // (1408 * 1920)
    encode.footprint(c, 896, 2)
    with pytest.raises(Exception):
        encode.footprint(c, 897, 2)


def test_the_footprint_is_linear_in_the_gops_and_grows_with_the_frames():
    c = _cfg(3840, 2160)
    b = [encode.footprint(c, n, 2)[0] for n in (1, 2, 3, 896)]
    assert abs((b[1] - b[0]) - (b[2] - b[1])) <= 65536  # (the composed walk's workspace is some forty arrays, each rounded up to 256 bytes)
    per_gop = b[1] - b[0]
    assert 200e6 < per_gop < 300e6  # two picture stores, original, input, maps, both CTU stores in the writer's form, the walk's state of 8 chains
    # (the walk's workspace is the composed walk's at every width since round 6 -- walk.hip; rounds 4-5: the fused kernel's up to 1024 chains)
    assert abs(b[3] - 896 * per_gop) < 0.01 * b[3]
    w = [encode.footprint(c, n, 2)[0] for n in (200, 300, 400)]
    assert abs((w[1] - w[0]) - (w[2] - w[1])) <= 65536 and abs((w[1] - w[0]) / 100 - per_gop) < 0.01 * per_gop
    assert encode.footprint(c, 16, 8)[0] > encode.footprint(c, 16, 2)[0]  # more frames to hold, more picture stores alive
    assert encode.footprint(_cfg(3840, 2160, threads=1), 16, 2)[0] < encode.footprint(c, 16, 2)[0]  # one chain: no second pass, no CTU stores


def test_a_job_is_cut_into_rounds_of_batches_that_fit():
    c = _cfg(3840, 2160)
    per_gop = encode.footprint(c, 2, 2)[0] - encode.footprint(c, 1, 2)[0]
    free = 286 * 10 ** 9
    rounds = encode.plan_batches(c, 3000, 2, free)
    flat = [b for r in rounds for b in r]
    assert [f for f, _ in flat] == [sum(n for _, n in flat[:i]) for i in range(len(flat))] and sum(n for _, n in flat) == 3000  # every GOP once, in order
    for r in rounds:
        assert len(r) <= 3 and all(1 <= n <= 896 for _, n in r)
        assert sum(n for _, n in r) * per_gop <= free - (8 << 30)
    assert rounds[0][0][1] == 896  # a round is filled as far as the limit and the memory go
    assert encode.plan_batches(c, 5, 2, free) == [[(0, 5)]]
    assert encode.plan_batches(c, 10, 2, free, max_batches=2, batch_gops=4) == [[(0, 4), (4, 4)], [(8, 2)]]
    with pytest.raises(Exception):
        encode.plan_batches(c, 5, 2, 1 << 20)


@pytest.mark.gpu
def test_gops_coded_in_batches_side_by_side_and_in_rounds_are_the_references(tmp_path):
    import _e2e
    import xeve_amd

    xeve_amd.init(0)
    w, h, gops, frames, seed, cli, threads = _enc.BATCH_CASES["gops_128x64_noise"]
    gold = _enc.golden()["batches"]["gops_128x64_noise"]["per_gop"]
    p = str(tmp_path / "in.yuv")
    _e2e.make_yuv(p, w, h, gops * frames, seed)
    data, fb = open(p, "rb").read(), w * h * 3 // 2 * frames
    c0 = _enc.config(w, h, cli, threads)
    cfg = encode.config(w, h, qp=c0.qp, keyint=c0.keyint, bframes=c0.bframes, closed_gop=c0.closed_gop, preset=c0.preset, threads=c0.threads, ref=c0.ref)
    N = 10  # GOP i = the case's GOP i % 3
    fed = []

    def feed(enc, first, n):
        fed.append((first, n))
        for g in range(n):
            enc.push_gop(g, data[((first + g) % gops) * fb:((first + g) % gops + 1) * fb])

    # four GOPs per batch, two batches at a time: a round of 4 + 4, then one of 2
    out = encode.encode_gops(cfg, N, frames, feed, max_batches=2, batch_gops=4)
    assert fed == [(0, 4), (4, 4), (8, 2)]
    assert [(len(o), _enc.md5(o)) for o in out] == [(gold[i % gops]["bytes"], gold[i % gops]["md5"]) for i in range(N)]
