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


def test_the_composed_walk_at_every_width_unless_pinned():
    """round 6: the composed walk with its side stream finishes a step sooner than the fused kernel at every width (walk.hip), so the choice by width is composed throughout"""
    assert ask() == [0] * 6
    assert ask(XEVE_HIP_WALK="auto") == [0] * 6
    assert ask(XEVE_HIP_WALK="") == [0] * 6
    assert ask(XEVE_HIP_WALK_AUTO_MAX="1024") == [1, 1, 1, 0, 0, 0]


def test_the_environment_pins_a_walk_or_moves_the_width():
    assert ask(XEVE_HIP_WALK="1") == [1] * 6
    assert ask(XEVE_HIP_WALK="0") == [0] * 6
    assert ask(XEVE_HIP_WALK_AUTO_MAX="4000") == [1, 1, 1, 1, 1, 0]
    assert ask(XEVE_HIP_WALK_AUTO_MAX="0") == [0] * 6


def test_the_choice_moves_at_run_time_and_comes_back():
    """xeve_hip_walk_select / xeve_hip_walk_team (encode.walk_select): what the GPU suite uses to run one process through both walks"""
    code = ("from xeve_amd import lib, encode; L = lib.load(); f = lambda: [L.xeve_hip_walk_fused(n) for n in (8, 5000)]; a = f()\n"
            "with encode.walk_select(0):\n b = f()\n with encode.walk_select(1, 3):\n  c = f(); t = L.xeve_hip_walk_team(-1)\n d = f()\n"
            "print(a, b, c, t, d, f(), L.xeve_hip_walk_team(99), L.xeve_hip_walk_select(7))")
    e = {k: v for k, v in os.environ.items() if not k.startswith("XEVE_HIP_WALK")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(e, PYTHONPATH=ROOT), timeout=120)
    assert p.returncode == 0, p.stderr[-800:]
    assert p.stdout.strip() == "[0, 0] [0, 0] [1, 1] 3 [0, 0] [0, 0] 0 -1", p.stdout
