"""Which CTU walk a batch gets (walk.hip xh_walk_enabled through xeve_hip_walk_fused): host logic only, no device.  The switch is read once per process, so every setting
is asked in a process of its own."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASK = "from xeve_amd import lib; L = lib.load(); print(' '.join(str(L.xeve_hip_walk_fused(n)) for n in (1, 8, 1024, 1025, 3584, 65535)))"


def ask(**env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("XEVE_HIP_WALK")}
    e.update(env, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", ASK], capture_output=True, text=True, env=e, timeout=120)
    assert p.returncode == 0, p.stderr[-800:]
    return [int(x) for x in p.stdout.split()]


def test_the_fused_walk_up_to_1024_chains_the_composed_walk_above():
    assert ask() == [1, 1, 1, 0, 0, 0]
    assert ask(XEVE_HIP_WALK="auto") == [1, 1, 1, 0, 0, 0]
    assert ask(XEVE_HIP_WALK="") == [1, 1, 1, 0, 0, 0]


def test_the_environment_pins_a_walk_or_moves_the_width():
    assert ask(XEVE_HIP_WALK="1") == [1] * 6
    assert ask(XEVE_HIP_WALK="0") == [0] * 6
    assert ask(XEVE_HIP_WALK_AUTO_MAX="4000") == [1, 1, 1, 1, 1, 0]
    assert ask(XEVE_HIP_WALK_AUTO_MAX="0") == [0] * 6
