#!/usr/bin/env python3
"""Debug aid: rerun one dumped fuzz case (tests/tools/fuzz_parity.py --dump) until the device's planes
differ from the oracle's and print where the first plane list diverges."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from buildingsegment_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402

d = np.load(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
xyz, nrm, neigh = d["xyz"], d["normals"], d["neigh"]
k = int(d["k"])
p = api.default_params(k=k)
p.th_thickness, p.th_point_count, p.cos_th = int(d["th"]), int(d["cnt"]), float(d["cos"])
opi, opl = O.region_grow(xyz, nrm, neigh, th_thickness=p.th_thickness, th_point_count=p.th_point_count, cos_th=p.cos_th)
ctx = api.Context(0)
for r in range(reps):
    pi, planes = ctx.region_grow(xyz, nrm, neigh, p)
    bad = None
    for q, pl in enumerate(planes):
        lst = opl["point_idx"][opl["offset"][q]:opl["offset"][q + 1]]
        if len(lst) != len(pl.pointIdx) or not np.array_equal(lst, pl.pointIdx):
            bad = (q, lst, np.asarray(pl.pointIdx))
            break
    if bad is None:
        continue
    q, lo, ld = bad
    m = min(len(lo), len(ld))
    pos = int(np.flatnonzero(lo[:m] != ld[:m])[0]) if (lo[:m] != ld[:m]).any() else m
    print(f"run {r}: plane {q} seed {lo[0]} sizes oracle {len(lo)} device {len(ld)} first difference at {pos}; labels equal {np.array_equal(pi, opi)}")
    print("  oracle", lo[max(0, pos - 4):pos + 6], "\n  device", ld[max(0, pos - 4):pos + 6])
    for name, a, b in (("oracle", lo, ld), ("device", ld, lo)):
        x = a[pos]
        w = np.flatnonzero(b == x)
        par = [int(t) for t in a[:pos] if x in neigh[t][1:]]
        print(f"  {name}'s point {x}: position in the other list {w}, label oracle {opi[x]} device {pi[x]}, parents before pos {par[:6]}")
    # who are the neighbours of the parent whose row is being processed?
    same = sorted(set(lo.tolist())) == sorted(set(ld.tolist()))
    print("  same point set:", same, " seeds of planes:", [int(opl['point_idx'][o]) for o in opl['offset'][:-1]][max(0, q - 3):q + 3])
    print("  timings", ctx.timings())
    break
else:
    print("no mismatch in", reps, "runs")
