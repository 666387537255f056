#!/usr/bin/env python3
"""Developer probe: microseconds per growth step on single-plane clouds of different sizes (records of 128 B: a
20 k-point plane fits one XCD's 4 MB L2, a 250 k-point plane only the Infinity Cache).  Run with BS_DEBUG=1; the
library prints '[bs] launch span ... us/step' per round."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from buildingsegment_amd import api, synth  # noqa: E402

ctx = api.Context(0)
for side in (140, 300, 500):
    pts = synth._face((0, 0, 0), (1, 0, 0), (0, 1, 0), side, side, 50, 7, 1)
    xyz = synth._finish(pts, 7, True)
    p = api.default_params(k=16)
    sys.stderr.write(f"[probe] plane {side}x{side} = {len(xyz)} points\n")
    sys.stderr.flush()
    ctx.segment(xyz, p)
    t = ctx.timings()
    sys.stderr.write(f"[probe]   grow_kernel_ms {t['grow_kernel_ms']:.2f} largest {t['largest_plane']} rounds {t['rg_rounds']}\n")
ctx.close()
