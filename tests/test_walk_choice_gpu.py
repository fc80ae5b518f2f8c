"""Both CTU walks stay pinned on the GPU whatever the library's own choice is (walk.hip: since round 6 the composed walk at every width; the fused kernel up to 1024
chains in rounds 4-5): reference-bitstream cases of the batch encoder once more with the composed walk pinned (the bench's path at width) and with the fused
kernel carrying three chains per team (teams of several chains are otherwise only formed beyond ~1000 chains).  In process (xeve_hip_walk_select / _team; round 4 spawned an
interpreter per pin); the CTU-tree tests take the same three settings through tests/conftest.py `each_walk`, the real-size cases through tests/test_enc_gpu.py."""
import pytest

import _e2e
import _enc
import test_enc_gpu as T

pytestmark = pytest.mark.gpu

SINGLE = ["tiny_ldb_fast", "tiny_ra_medium", "moving_ldb_ref3", "jumpy_ldb_fast"]
BATCHES = ["gops_128x64_noise", "gops_cif_noise_m8"]


@pytest.fixture(scope="module")
def hip():
    import xeve_amd
    from xeve_amd import encode

    xeve_amd.init(0)
    return encode


@pytest.fixture(scope="module")
def yuv_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("walk_choice_yuv")


@pytest.mark.parametrize("pin", ["composed", "composed_one_stream", "fused_3_chains_per_team"])
def test_the_encoder_with_the_walk_pinned(pin, hip, yuv_dir):
    from conftest import WALKS

    mode, team, side = WALKS[pin]
    L = __import__("xeve_amd.lib", fromlist=["load"]).load()
    with hip.walk_select(mode, team, side):
        assert L.xeve_hip_walk_fused(8) == mode and L.xeve_hip_walk_fused(100000) == mode  # (pinned: whatever the width)
        for name in SINGLE:
            w, h, n, seed, cli = _e2e.CASES[name]
            out, _ = T._run(hip, T._cfg(hip, w, h, cli), [T._frames(yuv_dir, name, w, h, n, seed)], n)
            assert (len(out[0]), _enc.md5(out[0])) == (T.E2E[name]["bytes"], T.E2E[name]["md5"]), (pin, name)
        for name in BATCHES:
            w, h, gops, frames, seed, cli, threads = _enc.BATCH_CASES[name]
            g = _enc.golden()["batches"][name]
            data, fb = T._frames(yuv_dir, name, w, h, gops * frames, seed), w * h * 3 // 2 * frames
            outs, _ = T._run(hip, T._cfg(hip, w, h, cli, threads), [data[i * fb:(i + 1) * fb] for i in range(gops)], frames)
            assert [(len(o), _enc.md5(o)) for o in outs] == [(p["bytes"], p["md5"]) for p in g["per_gop"]], (pin, name)
    assert L.xeve_hip_walk_fused(8) == 0 and L.xeve_hip_walk_fused(100000) == 0  # back to the library's choice (round 6: the composed walk at every width)
