#!/usr/bin/env python3
"""The speculative scheduler under repetition: N passes over one workload with the replay certificate on; every pass must
report 0 mismatches and the same label digest.  usage: audit_repeat.py [workload] [passes]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "urban_50m"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
xyz, k = bench.make_cloud(wl)
n = len(xyz)
dev = torch.device("cuda", 0)
ctx = api.Context(0)
ctx.set_audit(True)
p = api.default_params(k=k)
d_xyz = torch.from_numpy(xyz).to(dev)
d_lab = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
first = None
bad = 0
rej = {}
for it in range(passes):
    ctx.segment_dev(d_xyz.data_ptr(), n, d_lab.data_ptr(), p)
    t = ctx.timings()
    dig = hashlib.sha256(d_lab.cpu().numpy().tobytes()).hexdigest()[:16]
    first = first or dig
    ok = t["audit_mismatches"] == 0 and dig == first
    bad += 0 if ok else 1
    for key in ("rej_v1_robbed", "rej_v1_tag", "rej_v1_dup", "rej_v3_state"):
        rej[key] = rej.get(key, 0) + t[key]
    print(f"pass {it}: audit {t['audit_attempts']} attempts, {t['audit_mismatches']} mismatches, {t['audit_ms']:.0f} ms; rounds {t['rg_rounds']}; "
          f"labels {dig} {'ok' if ok else 'DIFFERENT'}", flush=True)
print("refusals over all passes:", rej)
print("OK" if not bad else "FAILED")
ctx.close()
sys.exit(1 if bad else 0)
