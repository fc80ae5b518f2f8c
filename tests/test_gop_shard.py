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
