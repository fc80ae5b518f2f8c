#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs oracle/_ref/xeveb_app = the unmodified reference compiled in place from /root/reference by oracle/Makefile).
Golden of BASELINE config 4 as a FULL closed GOP: 3840x2160, 8 frames of the seed-4 uniform 8-bit clip, `--preset medium --closed-gop -I 8 -m 8`.
Written to tests/golden/cfg4_8f_v1.json: the whole file's size + md5, and for every coded picture (coding order) the size + md5 of the bitstream UP TO AND INCLUDING that
picture's access unit -- bench.py stops its job after a few pictures and checks what GOP 0 has produced so far against these prefixes.  Also 1920x1080 (seed 3)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import reference_noise  # noqa: E402


def prefixes(evc):
    """(bytes, md5) of the stream up to the end of every slice NAL unit: a NAL unit = 4-byte big-endian length + payload; nal_unit_type = ((payload[0] >> 1) & 0x3F) - 1
    (xeve_eco_nalu, src_base/xeve_eco.c), slice types NONIDR 0 / IDR 1"""
    out, at = [], 0
    while at + 4 <= len(evc):
        n = int.from_bytes(evc[at:at + 4], "big")
        t = ((evc[at + 4] >> 1) & 0x3F) - 1
        at += 4 + n
        if t in (0, 1):
            out.append({"bytes": at, "md5": hashlib.md5(evc[:at]).hexdigest(), "idr": int(t == 1)})
    assert at == len(evc)
    return out


def run(w, h, frames, seed, name, out, preset="medium"):
    exe = os.path.join(ROOT, "oracle", "_ref", "xeveb_app")
    with tempfile.TemporaryDirectory() as d:
        yuv, evc = os.path.join(d, "in.yuv"), os.path.join(d, "o.evc")
        reference_noise(w * h * 3 // 2 * frames, seed).tofile(yuv)
        cli = ["--preset", preset, "--closed-gop", "-I", "8", "-m", "8"]
        t = time.time()
        subprocess.run([exe, "-i", yuv, "-w", str(w), "-h", str(h), "-z", "30", "--frames", str(frames), "-o", evc] + cli, check=True, stdout=subprocess.DEVNULL)
        wall = time.time() - t
        b = open(evc, "rb").read()
    out[name] = {"w": w, "h": h, "frames": frames, "seed": seed, "cli": cli, "bytes": len(b), "md5": hashlib.md5(b).hexdigest(), "after_picture": prefixes(b),
                 "reference_wall_s_in_the_build_container": round(wall, 1)}
    assert len(out[name]["after_picture"]) == frames


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "cfg4_8f_v1.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    # (round 5: the same two clips at presets slow and placebo -- bench.py --preset; the application's wall time in the build container is kept beside them)
    for (w, h, seed, name, preset) in ((1920, 1080, 3, "cfg3_1080p_closedgop_medium_8f_m8", "medium"), (3840, 2160, 4, "cfg4_2160p_closedgop_medium_8f_m8", "medium"),
                                       (1920, 1080, 3, "cfg3_1080p_closedgop_slow_8f_m8", "slow"), (3840, 2160, 4, "cfg4_2160p_closedgop_slow_8f_m8", "slow"),
                                       (1920, 1080, 3, "cfg3_1080p_closedgop_placebo_8f_m8", "placebo"), (3840, 2160, 4, "cfg4_2160p_closedgop_placebo_8f_m8", "placebo")):
        if name not in out:
            run(w, h, 8, seed, name, out, preset)
            json.dump(out, open(path, "w"), indent=1)
            print(name, out[name]["bytes"], out[name]["md5"], flush=True)
