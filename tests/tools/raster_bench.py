#!/usr/bin/env python3
"""Times the device-resident 2-D raster (bs_grid_picture_dev) on an urban cloud and
the CPU oracle beside it.  usage: python tests/tools/raster_bench.py [n_points] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.zeros(1, device="cuda")
from buildingsegment_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
xyz = synth.shift_to_origin(synth.urban(n, seed=3))
ext = xyz.max(0).astype(np.int32)
w, h = api.grid_dims(ext)
ctx = api.Context(0)
d_xyz = torch.from_numpy(xyz).cuda()
d_img = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
ctx.grid_picture_dev(d_xyz.data_ptr(), len(xyz), ext, d_img.data_ptr())
t = time.time()
for _ in range(reps):
    ctx.grid_picture_dev(d_xyz.data_ptr(), len(xyz), ext, d_img.data_ptr())
gpu_ms = (time.time() - t) / reps * 1e3
t = time.time()
oimg, _ = O.grid_picture(xyz, extent=ext)
cpu_ms = (time.time() - t) * 1e3
print(f"raster n={len(xyz)} image={w}x{h} gpu {gpu_ms:.2f} ms ({len(xyz) / gpu_ms / 1e3:.0f} Mpoints/s) "
      f"cpu oracle {cpu_ms:.0f} ms equal={np.array_equal(d_img.cpu().numpy(), oimg)}")
