#!/usr/bin/env python3
"""Full-size golden DIGESTS of the BASELINE.json workloads (build container; CPU only).

Runs the CPU oracle (oracle/bs_oracle.c) over a whole bench workload and stores
sha256 digests of everything the hot path returns -- neighbour indices, normal
bit patterns, labels, every plane's pointIdx list (concatenated, in commit order),
CSR offsets, plane normals and centres -- plus a few small summaries that localise
a mismatch (per-block label digests, plane sizes).  tests/test_gpu_fullsize.py
compares the HIP path's outputs with them at full size on the GPU box, where the
oracle itself would take minutes (10 M) to a quarter of an hour (50 M).

Stage 3 of the oracle is pinned to the reference compiled verbatim
(tests/test_oracle_golden.py); stages 1-2 are the restatement of Open3D 0.19
("parity unpinned").  Reference sites: tmc3/TMC3.cpp:213-217.

A workload may be given as NAME@kK: the same cloud searched with K neighbours -- the reference's own literals
(K = 15, r = 100, max_nn = 50; TMC3.cpp:215-216, my_function.h:63) are stored as facade_1m_k15 / urban_10m_k15.

`tie_rows` = queries whose k-list has an equal-d^2 pair inside it or at its boundary (k-th vs (k+1)-th neighbour):
exactly the rows where nanoflann's traversal order (the reference) may differ from the canonical ascending-index
order of this build -- the GPU path counts the same thing (bs_timings.tie_rows).

usage: make_digests.py [workload[@kK] ...]      (default: facade_1m urban_10m urban_50m)
"""
from __future__ import annotations

import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "digests.json")
BLOCK = 1 << 20  # label digests per block of 2^20 points

_XYZ = None
_K = 0


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def block_digests(a: np.ndarray, block: int = BLOCK) -> list[str]:
    """Short digests of consecutive row blocks (localises a mismatch)."""
    return [sha(a[i:i + block])[:16] for i in range(0, len(a), block)]


def _chunk(args):
    from oracle import oracle as O
    q0, q1 = args
    # k + 1 neighbours: the first k ARE the k-list (canonical order), the extra one decides boundary ties
    ng, nr = O.knn_normals(_XYZ, k=_K + 1, q0=q0, q1=q1)
    q = _XYZ[q0:q1].astype(np.int64)
    d = _XYZ[ng].astype(np.int64) - q[:, None, :]
    d2 = (d * d).sum(2)
    ties = int((np.diff(d2, axis=1) == 0).any(1).sum())
    return q0, q1, np.ascontiguousarray(ng[:, :_K]), nr, ties


def digest_outputs(neigh, normals, plane_idx, planes) -> dict:
    """planes: dict(id, normal, center, offset, point_idx) as oracle.region_grow returns it."""
    off = np.asarray(planes["offset"], np.int64)
    return {
        "neigh": sha(neigh),
        "normals_bits": sha(normals),
        "plane_idx": sha(plane_idx),
        "plane_idx_blocks": block_digests(plane_idx),
        "neigh_blocks": block_digests(neigh),
        "n_planes": int(len(planes["id"])),
        "labelled": int((plane_idx > 0).sum()),
        "point_idx": sha(np.asarray(planes["point_idx"], np.int32)),
        "offset": sha(off),
        "plane_normal_bits": sha(np.asarray(planes["normal"], np.float64)),
        "plane_center": sha(np.asarray(planes["center"], np.int32)),
        "plane_sizes_head": [int(v) for v in np.diff(off)[:64]],
        "largest_plane": int(np.diff(off).max()) if len(off) > 1 else 0,
    }


def run(workload: str, procs: int, k_override: int = 0) -> dict:
    global _XYZ, _K
    import bench
    from oracle import oracle as O
    t0 = time.time()
    xyz, k = bench.make_cloud(workload, 0)
    if k_override:
        k = k_override
    n = len(xyz)
    print(f"[{workload}] n={n} k={k} cloud in {time.time() - t0:.1f}s", flush=True)
    _XYZ, _K = xyz, k
    neigh = np.empty((n, k), np.int32)
    normals = np.empty((n, 3), np.float64)
    per = max(250_000, (n + 4 * procs - 1) // (4 * procs))
    jobs = [(q, min(n, q + per)) for q in range(0, n, per)]
    t1 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        tie_rows = 0
        for q0, q1, ng, nr, ties in pool.imap_unordered(_chunk, jobs):
            neigh[q0:q1] = ng
            normals[q0:q1] = nr
            tie_rows += ties
            print(f"  knn+normals [{q0}, {q1}) done at {time.time() - t1:.0f}s", flush=True)
    t2 = time.time()
    plane_idx, planes = O.region_grow(xyz, normals, neigh)
    t3 = time.time()
    print(f"[{workload}] knn+normals {t2 - t1:.0f}s ({procs} procs), region grow {t3 - t2:.0f}s", flush=True)
    d = digest_outputs(neigh, normals, plane_idx, planes)
    d.update({"workload": workload, "n": n, "k": k, "xyz": sha(xyz), "tie_rows": tie_rows,
              "n_seed_attempts": int(planes["n_seed_attempts"]),
              "oracle_seconds": {"knn_normals_wall": round(t2 - t1, 1), "procs": procs, "region_grow": round(t3 - t2, 1)}})
    return d


def main():
    wls = sys.argv[1:] or ["facade_1m", "urban_10m", "urban_50m"]
    procs = int(os.environ.get("BS_DIGEST_PROCS", "7"))
    db = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for w in wls:
        name, _, kk = w.partition("@k")
        key = f"{name}_k{kk}" if kk else name
        db[key] = run(name, procs, int(kk) if kk else 0)
        w = key
        with open(OUT, "w") as f:
            json.dump(db, f, indent=1, sort_keys=True)
        print(f"[{w}] digests written", flush=True)


if __name__ == "__main__":
    main()
