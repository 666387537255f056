#!/usr/bin/env python3
"""Full-size exactness check of a bench workload against the CPU oracle (GPU box;
takes minutes: the oracle is single threaded).  usage: check_large.py <workload> [repeats of the growth]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402

wl = sys.argv[1]
xyz, k = bench.make_cloud(wl, 0)
print(wl, "n", len(xyz), "k", k, flush=True)
ctx = api.Context(0)
t = time.time()
neigh, normals, plane_idx, planes = ctx.segment(xyz, api.default_params(k=k))
print("gpu segment (host buffers)", round(time.time() - t, 2), "s", ctx.timings(), flush=True)
t = time.time()
n = len(xyz)
oneigh = np.empty((n, k), np.int32)
onormals = np.empty((n, 3), np.float64)
chunk = 4_000_000  # progress lines keep the GPU box's silence watchdog quiet
for q0 in range(0, n, chunk):
    q1 = min(n, q0 + chunk)
    oneigh[q0:q1], onormals[q0:q1] = O.knn_normals(xyz, k=k, q0=q0, q1=q1)
    print("  oracle knn+normals", q1, "/", n, round(time.time() - t, 1), "s", flush=True)
print("oracle knn+normals", round(time.time() - t, 1), "s", flush=True)
print("neigh equal", np.array_equal(neigh, oneigh), "normals equal", np.array_equal(normals, onormals), flush=True)
t = time.time()
opi, opl = O.region_grow(xyz, onormals, oneigh)
print("oracle region grow", round(time.time() - t, 1), "s", flush=True)
ok = np.array_equal(plane_idx, opi) and len(planes) == len(opl["id"])
if ok and planes:
    ok = np.array_equal(np.concatenate([q.pointIdx for q in planes]), opl["point_idx"]) and \
        np.array_equal(np.stack([q.normal for q in planes]), opl["normal"]) and \
        np.array_equal(np.stack([q.center for q in planes]), opl["center"])
print("labels/planes equal", ok, "planes", len(planes), "labelled", int((plane_idx > 0).sum()), flush=True)
# the speculative grower's schedule is timing dependent: repeat the growth (argv[2] times) against
# the same oracle result
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nbad = 0
for r in range(reps):
    pi, pls = ctx.region_grow(xyz, onormals, oneigh, api.default_params(k=k))
    okr = np.array_equal(pi, opi) and len(pls) == len(opl["id"])
    if okr and pls:
        okr = np.array_equal(np.concatenate([q.pointIdx for q in pls]), opl["point_idx"]) and \
            np.array_equal(np.stack([q.normal for q in pls]), opl["normal"]) and \
            np.array_equal(np.stack([q.center for q in pls]), opl["center"])
    nbad += not okr
    if (r + 1) % 5 == 0 or not okr:
        print(f"  repeated growth {r + 1}/{reps}: mismatching runs so far {nbad}", flush=True)
if reps:
    print("repeated growth:", reps, "runs,", nbad, "mismatches", flush=True)
sys.exit(0 if ok and nbad == 0 else 1)
