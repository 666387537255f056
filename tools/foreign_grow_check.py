#!/usr/bin/env python3
"""bs_region_grow_dev on FOREIGN buffers (what rank r of the sharded stage 3 calls) against the
fused pipeline's grow time on the same cloud.  Prints both; the foreign call builds its own Morton order."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "urban_10m"
xyz, k = bench.make_cloud(name)
n = len(xyz)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
ctx = api.Context(0)
p = api.default_params(k=k)
d_xyz = torch.from_numpy(xyz).to(dev)
d_ng = torch.empty((n, k), dtype=torch.int32, device=dev)
d_nr = torch.empty((n, 3), dtype=torch.float64, device=dev)
d_l1 = torch.empty(n, dtype=torch.int32, device=dev)
d_l2 = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
res = {"workload": name, "n": n, "k": k, "fused": [], "foreign": []}
for _ in range(3):
    ctx.segment_dev(d_xyz.data_ptr(), n, d_l1.data_ptr(), p, d_ng.data_ptr(), d_nr.data_ptr())
    t = ctx.timings()
    res["fused"].append({kk: round(t[kk], 2) for kk in ("grow_ms", "grow_setup_ms", "grow_kernel_ms")})
# foreign: copies of the buffers (new pointers, nothing cached for them)
f_xyz, f_ng, f_nr = d_xyz.clone(), d_ng.clone(), d_nr.clone()
torch.cuda.synchronize()
for _ in range(3):
    ctx.region_grow_dev(f_xyz.data_ptr(), f_nr.data_ptr(), f_ng.data_ptr(), n, d_l2.data_ptr(), p)
    t = ctx.timings()
    res["foreign"].append({kk: round(t[kk], 2) for kk in ("grow_ms", "grow_setup_ms", "grow_kernel_ms")})
res["labels_equal"] = bool(torch.equal(d_l1, d_l2))
print(json.dumps(res))
