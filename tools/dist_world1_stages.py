#!/usr/bin/env python3
"""Stage times of dist.segment_sharded_dev at world 1 with the collectives forced through RCCL (what every rank of
`bench.py --gpus N` executes besides waiting for its peers).  usage: dist_world1_stages.py [workload] [passes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402
from buildingsegment_amd import dist as D  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "urban_50m"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
D.FORCE_COLLECTIVES = True
xyz, k = bench.make_cloud(wl) if hasattr(bench, "make_cloud") else (None, None)
if xyz is None:
    raise SystemExit("bench.make_cloud not found")
dev = torch.device("cuda", 0)
ctx = api.Context(0)
p = api.default_params(k=k)
d_xyz = torch.from_numpy(xyz).to(dev)
d_g = torch.arange(len(xyz), dtype=torch.int32, device=dev)
for it in range(passes):
    torch.cuda.synchronize()
    labels, info = D.segment_sharded_dev(ctx, d_xyz, d_g, len(xyz), p)
    torch.cuda.synchronize()
    st = info["stage_ms"]
    print(f"pass {it}: total {sum(st.values()):.1f} ms  " + "  ".join(f"{a[:-3]} {b:.1f}" for a, b in st.items()), flush=True)
ctx.close()
dist.destroy_process_group()
