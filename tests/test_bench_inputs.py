"""bench.py's input for GOP 0 must be the very clip whose reference bitstream is the committed golden (tests/golden/e2e_v1.json cfg4_*): SURVEY.md 8(d)'s recipe
random.seed(S); bytes(random.getrandbits(8) ...), which bench.py reproduces without the Python loop."""
import random

import numpy as np


def test_reference_noise_is_the_python_loop():
    import bench

    for seed, n in ((4, 50000), (1234, 7777)):
        random.seed(seed)
        want = bytes(random.getrandbits(8) for _ in range(n))
        assert bench.reference_noise(n, seed).tobytes() == want
    a = bench.reference_noise(4096, 4)
    assert a.dtype == np.uint8 and a.tobytes() == bench.reference_noise(8192, 4)[:4096].tobytes()


def test_golden_prefixes_of_the_full_gops():
    """tests/golden/cfg4_8f_v1.json (what bench.py checks a bounded job against): one prefix per coded picture, growing, the IDR first, the last one the whole file"""
    import hashlib
    import json
    import os

    import bench

    g = json.load(open(os.path.join(bench.ROOT, "tests", "golden", "cfg4_8f_v1.json")))
    assert {"cfg3_1080p_closedgop_medium_8f_m8", "cfg4_2160p_closedgop_medium_8f_m8"} <= set(g)  # (+ the same clips at presets slow / placebo: bench.py --preset)
    for name, r in g.items():
        p = r["after_picture"]
        assert len(p) == r["frames"] == 8 and [q["idr"] for q in p] == [1] + [0] * 7
        assert all(a["bytes"] < b["bytes"] for a, b in zip(p, p[1:])) and (p[-1]["bytes"], p[-1]["md5"]) == (r["bytes"], r["md5"])
        preset = r["cli"][1]
        assert name.split("_")[3] == preset and bench.golden_prefix(r["w"], r["h"], 8, 8, preset)[0] == name and bench.golden_prefix(r["w"], r["h"], 8, 1, preset) == (None, None)
        fake = {"after_picture": [{"bytes": 3, "md5": hashlib.md5(b"abc").hexdigest()}, {"bytes": 5, "md5": hashlib.md5(b"abcde").hexdigest()}]}
        assert bench.check_prefix(b"abc", fake) == 1 and bench.check_prefix(b"abcde", fake) == 2 and bench.check_prefix(b"abd", fake) == 0 and bench.check_prefix(b"abcd", fake) == 0


def test_the_cpu_baseline_counts_the_cores_the_container_grants():
    import bench

    h = bench.host_info()
    assert 1 <= h["cores_available"] <= h["usable_cores"] and (h["cgroup_cpu_quota"] is None or h["cores_available"] <= h["cgroup_cpu_quota"])
