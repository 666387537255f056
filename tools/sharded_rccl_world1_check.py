#!/usr/bin/env python3
"""bs_segment_sharded at world 1 through bs_comm_rccl with the collectives forced (BS_SHARD_FORCE_COMM=1) on a cloud
of tens of millions of points: the RCCL calls themselves (all-reduce of the n-entry parent array, grouped send/recv
of gigabytes) against the one-context pipeline.  usage: sharded_rccl_world1_check.py [workload]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BS_SHARD_FORCE_COMM"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from buildingsegment_amd import _lib, api  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "urban_50m"
xyz, k = bench.make_cloud(wl)
n = len(xyz)
torch.zeros(1, device="cuda:0")
L = _lib.load()
ctx = api.Context(0)
p = api.default_params(k=k)
_, _, want, planes0 = ctx.segment(xyz, p)
uid = C.create_string_buffer(128)
assert L.bs_comm_rccl_unique_id(uid) == 0
comm = C.c_void_p()
assert L.bs_comm_rccl_init(ctx._h, uid, 0, 1, C.byref(comm)) == 0
ops = _lib.CommOps()
assert L.bs_comm_rccl(comm, 0, 1, C.byref(ops)) == 0
dev = torch.device("cuda", 0)
d_xyz = torch.from_numpy(xyz).to(dev)
d_g = torch.arange(n, dtype=torch.int32, device=dev)
d_lab = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
bad = 0
for it in range(2):
    t = time.time()
    info = ctx.segment_sharded(ops, d_xyz.data_ptr(), d_g.data_ptr(), n, n, d_lab.data_ptr(), p)
    dt = time.time() - t
    same = bool(np.array_equal(d_lab.cpu().numpy(), want))
    npl = len(ctx.sharded_planes_fetch())
    bad += 0 if same and npl == len(planes0) else 1
    print(f"[rccl w1] pass {it}: labels equal {same}, planes {npl} (one context {len(planes0)}), components {info['components']}, {dt * 1e3:.0f} ms "
          f"(partition {info['ms_partition']:.0f} halo {info['ms_halo']:.0f} knn {info['ms_knn']:.0f} components {info['ms_components']:.0f} "
          f"redistribute {info['ms_redistribute']:.0f} grow {info['ms_grow']:.0f} labels {info['ms_labels']:.0f})", flush=True)
assert L.bs_comm_rccl_destroy(comm) == 0
ctx.close()
print("[rccl w1] OK" if not bad else "[rccl w1] FAILED")
sys.exit(1 if bad else 0)
