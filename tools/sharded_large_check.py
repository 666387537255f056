#!/usr/bin/env python3
"""One cloud of a few million points through bs_segment_sharded with thread-ranks on ONE GPU (bs_comm_local_create),
every rank holding an interleaved share of the points, against the one-context pipeline on the same GPU (whose
output the digest tests pin to the oracle).  usage: sharded_large_check.py [n_points] [world] [k]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from buildingsegment_amd import _lib, api, synth  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
xyz = synth.urban(n_req, seed=9)
n = len(xyz)
p = api.default_params(k=k)
ctx0 = api.Context(0)
t0 = time.time()
_, _, want, planes0 = ctx0.segment(xyz, p)
print(f"[check] {n} points, one context: {len(planes0)} planes, {time.time() - t0:.2f} s (with transfers)", flush=True)
ctx0.close()
L = _lib.load()
dev = torch.device("cuda", 0)
ops = (_lib.CommOps * world)()
assert L.bs_comm_local_create(world, ops) == 0
res = [None] * world


def run(r):
    try:
        ctx = api.Context(0)
        idx = np.arange(r, n, world, dtype=np.int32)  # an interleaved share: nothing spatial about it
        d_xyz = torch.from_numpy(np.ascontiguousarray(xyz[idx])).to(dev)
        d_g = torch.from_numpy(idx).to(dev)
        d_lab = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t = time.time()
        info = ctx.segment_sharded(ops[r], d_xyz.data_ptr(), d_g.data_ptr(), len(idx), n, d_lab.data_ptr(), p, halo=0.0)
        res[r] = (d_lab.cpu().numpy(), info, len(ctx.sharded_planes_fetch()), time.time() - t)
        ctx.close()
    except Exception as e:  # noqa: BLE001
        res[r] = e


ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in ts:
    t.start()
for t in ts:
    t.join()
for r in range(world):
    L.bs_comm_local_destroy(C.byref(ops[r]))
bad = 0
for r in range(world):
    if isinstance(res[r], Exception):
        print(f"[check] rank {r}: {res[r]!r}")
        bad += 1
        continue
    lab, info, npl, dt = res[r]
    same = bool(np.array_equal(lab, want))
    bad += 0 if same else 1
    print(f"[check] rank {r}: labels equal {same}; planes here {npl}; n_own {info['n_own']} n_local {info['n_local']} n_grow {info['n_grow']} "
          f"components {info['components']} cc_iterations {info['cc_iterations']} halo_retries {info['halo_retries']} {dt:.2f} s "
          f"(partition {info['ms_partition']:.0f} halo {info['ms_halo']:.0f} knn {info['ms_knn']:.0f} components {info['ms_components']:.0f} "
          f"redistribute {info['ms_redistribute']:.0f} grow {info['ms_grow']:.0f} labels {info['ms_labels']:.0f} ms)", flush=True)
if not bad:
    total = sum(r[2] for r in res)
    print(f"[check] planes over the ranks {total} (one context: {len(planes0)})")
    bad += 0 if total == len(planes0) else 1
print("[check] OK" if not bad else "[check] FAILED")
sys.exit(1 if bad else 0)
