"""CPU suite: closed-GOP sharding (the multi-GPU path) incl. a world_size-2 gloo run."""
import os
import socket
import subprocess
import sys

import pytest

from xeve_amd import gop

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_tiles_sequence_exactly_once():
    for total, k in [(64, 8), (17, 8), (8, 8), (1, 8), (0, 8), (100, 33)]:
        sh = gop.plan(total, k)
        assert sum(s.frames for s in sh) == total
        assert all(s.seek == i * k for i, s in enumerate(sh))
        assert all(0 < s.frames <= k for s in sh)


def test_rank_partition_is_disjoint_and_complete():
    for world in (1, 2, 4, 8):
        per = [gop.shards_for_rank(100, 8, r, world) for r in range(world)]
        order = gop.concat_order(per)
        assert [s.gop for s in order] == list(range(13))
        assert gop.app_args(order[12]) == ["--seek", "96", "--frames", "4"]


def test_bad_partition_detected():
    with pytest.raises(ValueError):
        gop.concat_order([gop.shards_for_rank(64, 8, 0, 2)])


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from xeve_amd import gop
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = gop.shards_for_rank(50, 8, r, w)
# stand-in for "encode my shards": each shard yields a deterministic byte string; only SIZES are exchanged,
# exactly like the real driver (bench.py) which exchanges nothing but timing -- the data path has no collective
sizes = torch.zeros(7, dtype=torch.int64)
for s in mine:
    sizes[s.gop] = 1000 + s.frames
dist.all_reduce(sizes)                     # control-plane only
t = torch.tensor([float(len(mine))]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if r == 0:
    assert sizes.tolist() == [1008] * 6 + [1002], sizes
    assert t.item() == 4.0
    print("OK", sizes.tolist())
dist.destroy_process_group()
"""


def test_two_rank_gloo_shards(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script), ROOT],
        capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


# ---- the shard driver on the batch encoder (gop.run_encoder_shards): two workers, a sequence whose last GOP is short ----------------------------------------------------
FAKE_WORKER = r"""
# a worker of gop.run_encoder_shards with the GPU engine swapped for the CPU harness (tests/_enc.py: the product's frame loop on the oracle) -- everything else,
# from the job file to the per-GOP part files, is the product's protocol
import json, os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import _enc
from xeve_amd import gop
spec = json.load(open(sys.argv[2]))
c = spec["config"]
cli = ["--preset", c["preset"], "--closed-gop", "-I", str(c["keyint"]), "-b", str(c["bframes"])]
cfg = _enc.config(c["w"], c["h"], cli, c.get("threads", 1))
fb = c["w"] * c["h"] * 3 // 2
data = open(spec["yuv"], "rb").read()
mine = gop.shards_for_rank(spec["total_frames"], spec["keyint"], spec["rank"], spec["world"])
by_len = {}
for s in mine:
    by_len.setdefault(s.frames, []).append(s)
for frames, group in by_len.items():
    outs = _enc.encode_cpu(cfg, [data[s.seek * fb:(s.seek + frames) * fb] for s in group], frames)
    for s, o in zip(group, outs):
        open(os.path.join(spec["dir"], "gop%06d.evc" % s.gop), "wb").write(o)
"""


def _closed_gop_case(tmp_path):
    import json

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _e2e

    w, h, n, seed, cli = _e2e.CASES["tiny_closed_gop"]  # 10 frames, --closed-gop -I 4 -b 1: GOPs of 4, 4 and 2 frames
    yuv = str(tmp_path / "in.yuv")
    _e2e.make_yuv(yuv, w, h, n, seed)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_v1.json")))["tiny_closed_gop"]
    return yuv, dict(w=w, h=h, qp=32, bframes=1, preset="fast", threads=1), n, 4, gold


def test_two_workers_code_their_gops_and_the_concatenation_is_the_references_one_run(tmp_path):
    """world size 2 on the CPU: worker 0 takes GOPs 0 and 2 (the short tail), worker 1 GOP 1; the joined file = the reference's single run over the ten frames"""
    import hashlib

    yuv, config, n, keyint, gold = _closed_gop_case(tmp_path)
    script = tmp_path / "fake_worker.py"
    script.write_text(FAKE_WORKER)
    out = str(tmp_path / "out.evc")
    r = gop.run_encoder_shards(yuv, out, config, n, keyint, devices=[0, 1], worker_cmd=[sys.executable, str(script), ROOT], timeout=600)
    b = open(out, "rb").read()
    assert (len(b), hashlib.md5(b).hexdigest()) == (gold["bytes"], gold["md5"]) and r["bytes"] == len(b) and sorted(w for w, _, _ in r["workers"]) == [0, 1]


def test_a_keyint_that_cuts_a_sub_gop_short_is_not_split_silently(tmp_path):
    """with -b 3 -I 5 the reference's own --seek / --frames runs do not join to its single run: the driver refuses unless told to split anyway"""
    yuv, config, n, keyint, _ = _closed_gop_case(tmp_path)
    with pytest.raises(ValueError, match="multiple of bframes"):
        gop.run_encoder_shards(yuv, str(tmp_path / "o.evc"), dict(config, bframes=3), n, 5, devices=[0], worker_cmd=["true"])


def test_a_failing_worker_fails_the_run(tmp_path):
    yuv, config, n, keyint, _ = _closed_gop_case(tmp_path)
    script = tmp_path / "bad_worker.py"
    script.write_text("import sys\nsys.stderr.write('no GPU here')\nsys.exit(3)\n")
    with pytest.raises(RuntimeError, match="no GPU here"):
        gop.run_encoder_shards(yuv, str(tmp_path / "o.evc"), config, n, keyint, devices=[0], worker_cmd=[sys.executable, str(script)], timeout=60)


@pytest.mark.gpu
def test_two_worker_processes_on_the_gpu_reproduce_the_references_one_run(tmp_path):
    """the real workers (xeve_amd.shard_worker on the batch encoder), two processes sharing GPU 0"""
    import hashlib

    yuv, config, n, keyint, gold = _closed_gop_case(tmp_path)
    out = str(tmp_path / "out.evc")
    gop.run_encoder_shards(yuv, out, config, n, keyint, devices=[0], per_device=2, timeout=600, env={"PYTHONPATH": ROOT})
    b = open(out, "rb").read()
    assert (len(b), hashlib.md5(b).hexdigest()) == (gold["bytes"], gold["md5"])


@pytest.mark.gpu
def test_bench_with_two_ranks_sharing_the_gpu():
    """the N > 1 control path of bench.py (rendezvous, barriers, max over ranks, rank 0's one JSON line) with two ranks on GPU 0 (XEVE_BENCH_SHARE_GPU=1:
    gloo, never used for numbers); every rank encodes its own GOPs, the aggregate counts both"""
    import json
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "256", "--height", "128", "--gops", "6", "--frames", "8", "--pictures", "3",
           "--batches", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, XEVE_BENCH_SHARE_GPU="1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 alone speaks
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["gops_in_lockstep"] == [6, 6] and r["config"]["pictures_run"] == 3
    assert r["bitstream_check"]["all_seeded_gops_same_bytes"] and r["bitstream_check"]["gop0_bytes_so_far"] > 0
    # both ranks' frames are in the aggregate: 2 ranks x 12 GOPs x the timed share (3 of 4 slices of 3 pictures)
    assert abs(r["config"]["frames_in_timed_region"] - 12 * 3 * 0.75) < 4.0
    assert r["roofline"]["launches_in_region"] > 0
