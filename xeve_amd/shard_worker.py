"""One worker of xeve_amd.gop.run_encoder_shards: codes the closed GOPs of ITS share of a sequence on the GPU it was given (HIP_VISIBLE_DEVICES: exactly one) and leaves
one bitstream file per GOP.  usage: python -m xeve_amd.shard_worker job.json"""
import json
import os
import sys


def main(argv):
    spec = json.load(open(argv[1]))
    import xeve_amd
    from xeve_amd import encode, gop

    xeve_amd.init(int(os.environ.get("XEVE_HIP_DEVICE", "0")))
    cfg = encode.config(**spec["config"])
    fb = cfg.w * cfg.h * 3 // 2 * (2 if cfg.reserved[1] > 8 else 1)
    mine = gop.shards_for_rank(spec["total_frames"], spec["keyint"], spec["rank"], spec["world"])
    by_len = {}
    for s in mine:  # (a batch's GOPs have one length: the sequence's last GOP may be shorter and then goes alone)
        by_len.setdefault(s.frames, []).append(s)
    import torch

    free = int(torch.cuda.mem_get_info()[0] * float(spec.get("memory_share", 1.0)))
    with open(spec["yuv"], "rb") as f:
        for frames, group in sorted(by_len.items(), reverse=True):

            def feed(enc, first, n, group=group, frames=frames):
                for j in range(n):
                    f.seek(group[first + j].seek * fb)
                    for k in range(frames):
                        enc.push(j, k, f.read(fb))

            for s, stream in zip(group, encode.encode_gops(cfg, len(group), frames, feed, free_bytes=free)):
                tmp = os.path.join(spec["dir"], "gop%06d.evc.part" % s.gop)
                with open(tmp, "wb") as o:
                    o.write(stream)
                os.replace(tmp, os.path.join(spec["dir"], "gop%06d.evc" % s.gop))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
