#!/usr/bin/env python3
"""Makes tests/golden/tree_v1.npz: CTUs of real encodes -- what the CTU mode decision (mode_analyze_lcu -> mode_coding_tree, src_base/xeve_mode.c:2007-2610) was handed
and what THE REFERENCE made of it -- recorded by the LD_PRELOAD adapter (oracle/ref_shadow.c, XEVE_SHIM_TREE_GOLDEN) inside the unmodified encoder compiled in place under
oracle/_ref.  Run here (needs /root/reference); the .npz holds numeric arrays only.  Reference pictures that several records share are stored once."""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _e2e import CASES, REF_APP, SHADOW, make_yuv  # noqa: E402
from _libs import ORACLE_SO  # noqa: E402

# clip, CTUs recorded of every picture
PLAN = [("noise_allintra_medium", "0"), ("moving_ldb_fast", "0,3"), ("moving_ra_medium", "0,1")]  # noise (every node decided), then drifting texture: skip / direct / uni- and bi-predicted CUs


def records(path):
    d, at, rec = open(path, "rb").read(), 0, {}
    while at < len(d):
        name = d[at:at + 16].split(b"\0")[0].decode()
        n = struct.unpack("<q", d[at + 16:at + 24])[0]
        body = d[at + 24:at + 24 + n]
        at += 24 + n
        if name == "end":
            yield rec
            rec = {}
        else:
            rec[name] = body


def main():
    out, planes, k = {}, {}, 0
    with tempfile.TemporaryDirectory() as tmp:
        for clip, ctus in PLAN:
            w, h, n, seed, extra = CASES[clip]
            yuv, dump = os.path.join(tmp, "in.yuv"), os.path.join(tmp, "dump.bin")
            make_yuv(yuv, w, h, n, seed)
            if os.path.exists(dump):
                os.remove(dump)
            cmd = [REF_APP, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(n), "-m", "1", "-v", "0", "-o", os.path.join(tmp, "o.evc")] + list(extra)
            env = dict(os.environ, LD_PRELOAD=SHADOW, XEVE_SHIM_SHADOW_TREE=ORACLE_SO, XEVE_SHIM_TREE_GOLDEN=dump, XEVE_SHIM_TREE_GOLDEN_CTUS=ctus, XEVE_SHIM_SHADOW_NO_PICTURE="1")
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            assert p.returncode == 0 and ", 0 differ" in p.stderr, p.stderr[-800:]
            recs = list(records(dump))
            writer = {tuple(np.frombuffer(r["wr_head"], np.int32)[[0, 4]]): r for r in recs if "wr_head" in r}  # (poc, lcu) -> the reference writer's side of that CTU
            for rec in recs:
                if "wr_head" in rec:
                    continue
                hd = np.frombuffer(rec["head"], np.int32)
                rec.update({k_: v for k_, v in writer.get((hd[0], hd[4]), {}).items()})  # (absent for the last CTU of a picture: the tile's end follows it)
                for name, body in rec.items():
                    if name.startswith("ref") and name[3].isdigit() and not name.endswith("_poc"):  # a reference plane: stored once per content
                        key = hashlib.md5(body).hexdigest()[:12]
                        planes.setdefault(key, np.frombuffer(body, np.int16).copy())
                        out["r%d_%s" % (k, name)] = np.frombuffer(key.encode(), np.uint8).copy()
                    else:
                        out["r%d_%s" % (k, name)] = np.frombuffer(body, np.uint8).copy()
                out["r%d_clip" % k] = np.frombuffer(clip.encode(), np.uint8).copy()
                k += 1
    for key, a in planes.items():
        out["plane_" + key] = a
    out["n_records"] = np.array([k], np.int32)
    path = os.path.join(HERE, "tree_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d records, %d unique reference planes, %.2f MB" % (path, k, len(planes), os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
