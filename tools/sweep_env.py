#!/usr/bin/env python3
"""Sweeps environment knobs of the region grower on ONE resident cloud (the cloud is generated
once; the library reads the variables at every call).
usage: python tools/sweep_env.py <workload> <reps> VAR=v1,v2,... [VAR2=...]   (cartesian product)"""
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.zeros(1, device="cuda")
import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402

wl, reps = sys.argv[1], int(sys.argv[2])
knobs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[3:]]
xyz, k = bench.make_cloud(wl, 0)
n = len(xyz)
ctx = api.Context(0)
p = api.default_params(k=k)
d_xyz = torch.from_numpy(xyz).cuda()
d_neigh = torch.empty((n, k), dtype=torch.int32, device="cuda")
d_nrm = torch.empty((n, 3), dtype=torch.float64, device="cuda")
d_pl = torch.empty((n,), dtype=torch.int32, device="cuda")
ctx.segment_dev(d_xyz.data_ptr(), n, d_pl.data_ptr(), p, d_neigh.data_ptr(), d_nrm.data_ptr())
for combo in itertools.product(*[v for _, v in knobs]):
    for (name, _), v in zip(knobs, combo):
        os.environ[name] = v
    res = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.segment_dev(d_xyz.data_ptr(), n, d_pl.data_ptr(), p, d_neigh.data_ptr(), d_nrm.data_ptr())
        torch.cuda.synchronize()
        tm = ctx.timings()
        res.append(((time.perf_counter() - t) * 1e3, tm["grow_kernel_ms"], tm["rg_rounds"]))
    print(" ".join(f"{a[0]}={v}" for a, v in zip(knobs, combo)), "| total ms", [round(r[0], 1) for r in res],
          "kernel ms", [round(r[1], 1) for r in res], "rounds", [r[2] for r in res], flush=True)
