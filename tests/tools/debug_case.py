import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from buildingsegment_amd import api
from oracle import oracle as O
g = np.load(sys.argv[1])
xyz, nr, ng = g["xyz"], g["normals"], g["neigh"]
k = int(g["k"])
p = api.default_params(k=k, th_thickness=int(g["th"]), th_point_count=int(g["cnt"]), cos_th=float(g["cos"]), rg_mode=2)
opi, opl = O.region_grow(xyz, nr, ng, th_thickness=p.th_thickness, th_point_count=p.th_point_count, cos_th=p.cos_th)
oseeds = [int(opl["point_idx"][o]) for o in opl["offset"][:-1]]
print("oracle planes", list(zip(oseeds, np.diff(opl["offset"]).tolist())))
ctx = api.Context(0)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    try:
        pi, planes = ctx.region_grow(xyz, nr, ng, p)
    except api.BsError as e:
        print("rep", rep, "ERROR", e)
        continue
    gs = [(int(q.pointIdx[0]), len(q.pointIdx)) for q in planes]
    print("rep", rep, "rounds", ctx.timings()["rg_rounds"], "labels differ", int((pi != opi).sum()), "planes", gs)
    if gs != list(zip(oseeds, np.diff(opl["offset"]).tolist())):
        bad = np.nonzero(pi != opi)[0]
        print("   first differing points", bad[:10], "gpu", pi[bad[:10]], "oracle", opi[bad[:10]])
