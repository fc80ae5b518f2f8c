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
