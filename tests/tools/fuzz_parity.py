#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the CPU oracle (GPU box only).

Random small clouds (surfaces, blobs, lattices with heavy ties), random k,
thresholds, normal noise and point orders; every stage is compared bit for bit
(neighbour indices, normals, labels, plane lists) in both region-grow modes.
usage: python tests/tools/fuzz_parity.py [--cases N] [--seed S] [--log FILE]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def make_case(rng):
    kind = rng.integers(0, 5)
    n = int(rng.integers(300, 30000))
    if kind == 0:  # noisy planes meeting at edges
        m = max(int(np.sqrt(n / 3)), 8)
        sp = int(rng.integers(20, 80))
        u, v = np.meshgrid(np.arange(m) * sp, np.arange(m) * sp, indexing="ij")
        j = int(rng.integers(0, sp // 2 + 1))
        u = u.ravel() + rng.integers(-j, j + 1, m * m)
        v = v.ravel() + rng.integers(-j, j + 1, m * m)
        t = int(rng.integers(0, 60))
        w = rng.integers(-t, t + 1, m * m)
        pts = np.concatenate([np.stack([u, v, w], 1), np.stack([u, w, v + 60], 1), np.stack([w, u + 60, v + 60], 1)])
    elif kind == 1:  # uniform blob
        L = int(50 * n ** (1 / 3) * rng.uniform(0.5, 3))
        pts = rng.integers(0, max(L, 4), (n, 3))
    elif kind == 2:  # lattice with duplicates and exact ties
        g = int(rng.integers(3, 14))
        pts = rng.integers(0, g, (n, 3)) * int(rng.integers(5, 60))
    elif kind == 3:  # curved sheet
        a = rng.uniform(0, 4000, n)
        b = rng.uniform(0, 4000, n)
        pts = np.stack([a, b, 300 * np.sin(a / 500.0) + rng.normal(0, 5, n)], 1)
    else:  # clusters far apart
        c = rng.integers(0, 200000, (int(rng.integers(2, 6)), 3))
        pts = c[rng.integers(0, len(c), n)] + rng.integers(-400, 400, (n, 3))
    pts = np.asarray(pts, dtype=np.int64)
    if rng.random() < 0.7:
        pts = pts[rng.permutation(len(pts))]
    if rng.random() < 0.3:
        pts = pts + rng.integers(-100000, 100000, 3)
    return np.ascontiguousarray(pts.astype(np.int32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default="")
    ap.add_argument("--dump", default="", help="directory for the inputs of failing cases")
    ap.add_argument("--only", type=int, default=-1, help="run just this case of the sequence (the others are only drawn)")
    ap.add_argument("--audit", action="store_true",
                    help="also switch on the replay certificate (bs_set_audit) and count its mismatches as failures")
    ap.add_argument("--repeat", type=int, default=1,
                    help="device runs per case and mode: exposes timing-dependent (nondeterministic) mismatches")
    args = ap.parse_args()
    from buildingsegment_amd import api
    from oracle import oracle as O
    ctx = api.Context(0)
    if args.audit:
        ctx.set_audit(True)
    rng = np.random.default_rng(args.seed)
    log = open(args.log, "a") if args.log else sys.stdout
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        xyz = make_case(rng)
        n = len(xyz)
        k = int(rng.integers(2, 33))
        if n < k:
            continue
        radius = float(rng.choice([30.0, 100.0, 100.0, 250.0]))
        max_nn = int(rng.choice([5, 50, 50, 64]))
        p = api.default_params(k=k, radius=radius, max_nn=max_nn, th_thickness=int(rng.choice([20, 300, 300, 2000])),
                               th_point_count=int(rng.choice([0, 5, 400, 400])),
                               cos_th=float(rng.choice([0.0, 0.5, 0.88, 0.88, 0.99])))
        if args.only >= 0 and case != args.only:  # keep the random stream in step, skip the work
            if rng.random() < 0.5:
                rng.normal(0, rng.choice([0.05, 0.3]), (n, 3))
            continue
        neigh, normals = ctx.knn_normals(xyz, p)
        oneigh, onormals = O.knn_normals(xyz, k=k, radius=radius, max_nn=max_nn)
        ok = np.array_equal(neigh, oneigh) and np.array_equal(normals, onormals)
        nrm = normals
        if rng.random() < 0.5:  # perturbed normals: stress the grower with orphans / small planes
            nrm = normals + rng.normal(0, rng.choice([0.05, 0.3]), normals.shape)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            nrm = np.ascontiguousarray(nrm)
        opi, opl = O.region_grow(xyz, nrm, oneigh, th_thickness=p.th_thickness, th_point_count=p.th_point_count,
                                 cos_th=p.cos_th)
        rounds = 0
        why = ""
        for mode in [2] * args.repeat + [1]:
            p.rg_mode = mode
            try:
                pi, planes = ctx.region_grow(xyz, nrm, oneigh, p)
            except api.BsError as e:
                ok = False
                why += f" mode{mode}:{e}"
                continue
            if mode == 2:
                rounds = ctx.timings()["rg_rounds"]
                tma = ctx.timings()
                if args.audit and (tma["audit_mismatches"] != 0 or tma["audit_attempts"] != tma["n_seed_attempts"]):
                    ok = False
                    why += f" audit:{tma['audit_mismatches']}_mismatches_of_{tma['audit_attempts']}_attempts(finalised:{tma['n_seed_attempts']})"
            okm = np.array_equal(pi, opi) and len(planes) == len(opl["id"])
            if okm and planes:
                okm = (np.array_equal(np.concatenate([q.pointIdx for q in planes]), opl["point_idx"])
                       and np.array_equal(np.stack([q.normal for q in planes]), opl["normal"])
                       and np.array_equal(np.stack([q.center for q in planes]), opl["center"]))
            if not okm:
                why += f" mode{mode}:labels_differ={int((pi != opi).sum())},planes={len(planes)}vs{len(opl['id'])}"
                if planes and len(planes) == len(opl["id"]):  # which plane field differs first
                    for q, (qq, on, oc) in enumerate(zip(planes, opl["normal"], opl["center"])):
                        lst = opl["point_idx"][opl["offset"][q]:opl["offset"][q + 1]]
                        if not (np.array_equal(qq.pointIdx, lst) and np.array_equal(qq.normal, on)
                                and np.array_equal(qq.center, oc)):
                            why += (f",plane{q}(n={len(lst)}):list={np.array_equal(qq.pointIdx, lst)}"
                                    f",normal={list(qq.normal)}vs{list(on)},center={list(qq.center)}vs{list(oc)}")
                            break
            ok = ok and okm
        if not ok and args.dump:
            np.savez_compressed(os.path.join(args.dump, f"fuzz_fail_{args.seed}_{case}.npz"), xyz=xyz, normals=nrm,
                                neigh=oneigh, k=k, th=p.th_thickness, cnt=p.th_point_count, cos=p.cos_th)
        bad += not ok
        print(f"case {case} n={n} k={k} r={radius} M={max_nn} th={p.th_thickness} cnt={p.th_point_count} "
              f"cos={p.cos_th} planes={len(opl['id'])} rounds={rounds} {'ok' if ok else 'MISMATCH' + why}", file=log, flush=True)
    print(f"done: {args.cases} cases, {bad} mismatches, {time.time() - t0:.1f}s", file=log, flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
