"""The grower's built-in self-checks (BS_VERIFY=1) stay silent on real runs.

BS_VERIFY makes bs_grow_spec.hip re-derive, with independent kernels, two things the fast path
maintains incrementally: the orphan-maker owner structure (is it a fixed point of its equations?)
and the per-point records (do they equal a full refresh after every round?).  A mismatch prints a
`[bs] VERIFY:` line on stderr.  The check runs in a child process because the switch is read from
the environment."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, glob, os
import numpy as np
sys.path.insert(0, sys.argv[1])
from buildingsegment_amd import api, synth
ctx = api.Context(0)
rounds = 0
for name, xyz, k in (("plane_cube", synth.plane_cube(), 15), ("facade", synth.facade(550, seed=3), 16),
                     ("urban", synth.urban(600_000, seed=5), 16)):
    p = api.default_params(k=k)
    neigh, normals, plane_idx, planes = ctx.segment(np.ascontiguousarray(xyz), p)
    rounds += ctx.timings()["rg_rounds"]
    assert int((plane_idx >= 0).sum()) > 0, name
print("rounds", rounds)
"""


def test_incremental_structures_match_their_definitions():
    env = dict(os.environ, BS_VERIFY="1")
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rounds" in out.stdout
    assert "VERIFY:" not in out.stderr, out.stderr
