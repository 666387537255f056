#!/usr/bin/env python3
"""Size-independent property check of a workload too large for the CPU oracle (GPU box):
segments the named bench workload through the host-buffer C ABI and runs the property checks of
tests/test_gpu_fullsize.py (k-list order, unit normals, label range, per-plane recomputation of centre and
normal from the returned list, list membership / duplicates).  usage: check_props_large.py <workload>"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from buildingsegment_amd import api  # noqa: E402

spec = importlib.util.spec_from_file_location("tf", os.path.join(ROOT, "tests", "test_gpu_fullsize.py"))
tf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tf)

wl = sys.argv[1]
t = time.time()
xyz, k = bench.make_cloud(wl)
print(wl, "n", len(xyz), "k", k, "cloud", round(time.time() - t, 1), "s", flush=True)
ctx = api.Context(0)
t = time.time()
neigh, normals, plane_idx, planes = ctx.segment(xyz, api.default_params(k=k))
tm = ctx.timings()
print("segment (host buffers)", round(time.time() - t, 1), "s; device", round(tm["total_ms"], 1), "ms =",
      round(len(xyz) / tm["total_ms"] / 1e3, 1), "Mpoints/s;", {kk: tm[kk] for kk in ("grid_ms", "knn_ms", "grow_ms", "grow_kernel_ms",
                                                                                 "grow_setup_ms", "rg_rounds", "largest_plane",
                                                                                 "validation_rejects")}, flush=True)
t = time.time()
tf.check_properties(xyz, k, neigh, normals, plane_idx, planes)
print("properties ok:", len(planes), "planes,", int((plane_idx > 0).sum()), "labelled,", round(time.time() - t, 1), "s", flush=True)
