"""Closed-GOP sharding of a sequence over GPUs / encoder processes.

With ``--closed-gop -I K`` every K-frame group starts with an IDR and is encoded independently of all
others (reference: src_base/xeve_enc.c:1083-1095,1154-1167,1980-1993); concatenating the per-group
bitstreams in order is byte-identical to the monolithic encode.  So the multi-GPU path needs NO
collective: rank r of `world` takes GOPs r, r + world, r + 2*world, ... and the host concatenates.
"""
from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class Shard:
    gop: int  # GOP index in display order
    seek: int  # first input frame       (xeve_app --seek, app/xeve_app.c:1171-1202)
    frames: int  # number of input frames  (xeve_app --frames)


def plan(total_frames: int, keyint: int) -> List[Shard]:
    if total_frames < 0 or keyint <= 0:
        raise ValueError("total_frames >= 0 and keyint > 0 required")
    return [Shard(g, s, min(keyint, total_frames - s)) for g, s in enumerate(range(0, total_frames, keyint))]


def shards_for_rank(total_frames: int, keyint: int, rank: int, world: int) -> List[Shard]:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [s for s in plan(total_frames, keyint) if s.gop % world == rank]


def app_args(shard: Shard) -> List[str]:
    """CLI fragment that makes the reference app encode exactly this shard."""
    return ["--seek", str(shard.seek), "--frames", str(shard.frames)]


def concat_order(per_rank: List[List[Shard]]) -> List[Shard]:
    """Order in which the host concatenates the shard bitstreams gathered from all ranks."""
    allsh = sorted((s for lst in per_rank for s in lst), key=lambda s: s.gop)
    if [s.gop for s in allsh] != list(range(len(allsh))):
        raise ValueError("shards do not tile the sequence exactly once")
    return allsh
