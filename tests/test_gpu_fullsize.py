"""Full-size parity of the BASELINE.json workloads on the GPU box (configs[1], configs[2] and the
north-star's 50 M cloud on one GPU; /root/reference/tmc3/TMC3.cpp:213-217 is the path):

  * digest comparison: sha256 of neighbour indices, normal bit patterns, labels and every
    plane's pointIdx list / centre / normal against tests/golden/digests.json, which
    tests/golden/make_digests.py produced by running the CPU oracle over the WHOLE workload in
    the build container (the oracle needs 70 s - 5 min of 6 cores there; here it would not fit
    the test budget).  Stage 3 of that oracle is pinned to the reference compiled verbatim,
    stages 1-2 are "parity unpinned" (DESIGN.md section 2).
  * size-independent properties checked from the HIP outputs alone: every k-list is sorted by
    (d^2, index) and starts with the query itself; normals are unit length with z >= 0; labels are
    -1 or in [1, n_planes + 1]; each plane's centre and normal RECOMPUTED from its returned
    pointIdx list (wrapping uint32 sum / size_t divide, f64 sum in list order: quirks Q3 and Q7)
    equal the returned ones bit for bit; every list member carries the plane's label, the seed
    may appear twice (quirk Q1) and nothing else may.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "digests.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def blocks(a, block=1 << 20):
    return [sha(a[i:i + block])[:16] for i in range(0, len(a), block)]


def check_properties(xyz, k, neigh, normals, plane_idx, planes, th_count=400):
    n = len(xyz)
    # --- k-lists (sample): self first, sorted by (d^2, index) ---
    rs = np.random.default_rng(5).choice(n, size=min(n, 200_000), replace=False)
    q = xyz[rs].astype(np.int64)
    d = xyz[neigh[rs]].astype(np.int64) - q[:, None, :]
    d2 = (d * d).sum(2)
    assert (neigh[rs, 0] == rs).all() or (d2[:, 0] == 0).all()
    assert (np.diff(d2, axis=1) >= 0).all()
    tie = np.diff(d2, axis=1) == 0
    assert (np.diff(neigh[rs].astype(np.int64), axis=1)[tie] > 0).all()  # equal d^2: ascending index
    assert neigh.min() >= 0 and neigh.max() < n
    # --- normals ---
    nn = (normals * normals).sum(1)
    assert np.abs(nn - 1.0).max() < 1e-12 and (normals[:, 2] >= 0).all()
    # --- labels ---
    npl = len(planes)
    assert plane_idx.min() >= -1 and plane_idx.max() <= npl + 1 and not (plane_idx == 0).any()
    # --- planes: recompute centre / normal from the list; membership ---
    for t, pl in enumerate(planes):
        li = pl.pointIdx
        assert pl.id == t + 1 and len(li) > th_count
        assert (plane_idx[li[1:]] == pl.id).all()
        uniq, cnt = np.unique(li, return_counts=True)
        dup = uniq[cnt > 1]
        assert len(dup) == 0 or (len(dup) == 1 and dup[0] == li[0] and cnt.max() == 2)  # only the seed, once more (Q1)
        csum = xyz[li].astype(np.uint32).sum(0, dtype=np.uint32).view(np.int32)  # wrapping int32 sum (Q3)
        centre = ((csum.astype(np.int64).view(np.uint64)) // np.uint64(len(li))).astype(np.uint32).view(np.int32)
        assert np.array_equal(centre, pl.center), (t, centre, pl.center)
        S = np.add.accumulate(normals[li], axis=0)[-1]  # sequential f64 sum in list order (Q7)
        nrm = np.sqrt((S[0] * S[0] + S[1] * S[1]) + S[2] * S[2])
        assert np.array_equal(S / nrm, pl.normal), (t, S / nrm, pl.normal)


# (NAME_k15: the same cloud with the reference's own literals K = 15, r = 100, max_nn = 50 -- TMC3.cpp:215-216)
@pytest.mark.parametrize("key", ["facade_1m", "urban_10m", "urban_50m", "facade_1m_k15", "urban_10m_k15"])
def test_fullsize_digests_and_properties(gpu_ctx, key):
    import bench
    from buildingsegment_amd import api
    gold = json.load(open(GOLD))[key]
    workload = gold["workload"]
    t0 = time.time()
    xyz, k = bench.make_cloud(workload)
    k = gold["k"]
    assert len(xyz) == gold["n"]
    assert sha(xyz) == gold["xyz"], "synthetic generator drifted: regenerate tests/golden/digests.json"
    print(f"\n[{workload}] cloud {time.time() - t0:.1f}s", flush=True)
    t0 = time.time()
    gpu_ctx.set_audit(True)  # replay of every plane attempt against the final owners (bs_set_audit)
    try:
        neigh, normals, plane_idx, planes = gpu_ctx.segment(xyz, api.default_params(k=k))
    finally:
        gpu_ctx.set_audit(False)
    tm = gpu_ctx.timings()
    assert tm["audit_mismatches"] == 0 and tm["audit_attempts"] == tm["n_seed_attempts"], tm
    print(f"[{workload}] audit: {tm['audit_attempts']} attempts replayed in {tm['audit_ms']:.0f} ms, 0 mismatches", flush=True)
    print(f"[{workload}] segment (host buffers) {time.time() - t0:.1f}s, device {tm['total_ms']:.0f} ms, "
          f"rounds {tm['rg_rounds']}, planes {len(planes)}", flush=True)
    # digests vs the CPU oracle's
    bad_n = [i for i, (a, b) in enumerate(zip(blocks(neigh), gold["neigh_blocks"])) if a != b]
    assert not bad_n, f"neighbour indices differ from the oracle in 2^20-point blocks {bad_n[:8]}"
    assert sha(neigh) == gold["neigh"]
    assert sha(normals) == gold["normals_bits"], "normal bit patterns differ from the oracle"
    bad_l = [i for i, (a, b) in enumerate(zip(blocks(plane_idx), gold["plane_idx_blocks"])) if a != b]
    assert not bad_l, f"labels differ from the oracle in 2^20-point blocks {bad_l[:8]}"
    assert sha(plane_idx) == gold["plane_idx"]
    assert len(planes) == gold["n_planes"] and int((plane_idx > 0).sum()) == gold["labelled"]
    sizes = [len(p.pointIdx) for p in planes]
    assert sizes[:64] == gold["plane_sizes_head"] and max(sizes) == gold["largest_plane"]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    assert sha(off) == gold["offset"]
    assert sha(np.concatenate([p.pointIdx for p in planes]).astype(np.int32)) == gold["point_idx"]
    assert sha(np.stack([p.normal for p in planes])) == gold["plane_normal_bits"]
    assert sha(np.stack([p.center for p in planes]).astype(np.int32)) == gold["plane_center"]
    assert tm["n_seed_attempts"] >= len(planes)
    if "tie_rows" in gold:  # exposure to the reference's kd-tree tie order: the same count as the CPU oracle's
        assert tm["tie_rows"] == gold["tie_rows"], (tm["tie_rows"], gold["tie_rows"])
        print(f"[{workload}] tie rows: {tm['tie_rows']} of {len(xyz)} ({100.0 * tm['tie_rows'] / len(xyz):.3f} %)", flush=True)
    t0 = time.time()
    check_properties(xyz, k, neigh, normals, plane_idx, planes)
    print(f"[{workload}] properties {time.time() - t0:.1f}s", flush=True)


def test_urban_200m_on_one_gpu_properties(gpu_ctx):
    """BASELINE.json configs[4]'s data size -- the 200 M-point multi-building scene (synth.urban seed 5, k=16) --
    through the whole path on ONE MI355X (it fits: ~90 GB of device buffers; n (k-1) = 3.0e9 reverse edges need
    the 64-bit offsets).  The CPU oracle would need hours at this size, so there are no digests: the
    size-independent properties are checked from the HIP outputs alone -- k-list order, unit normals, label
    range, list membership / duplicates, and every plane's centre and normal recomputed from its returned
    list (bit for bit).  Takes ~1 min of host time for the cloud."""
    import bench
    from buildingsegment_amd import api
    t0 = time.time()
    xyz, k = bench.make_cloud("urban_200m")
    assert len(xyz) == 200_000_000 and k == 16
    print(f"\n[urban_200m] cloud {time.time() - t0:.1f}s", flush=True)
    t0 = time.time()
    gpu_ctx.set_audit(True)  # no oracle at this size: the replay certificate stands in for the digests
    try:
        neigh, normals, plane_idx, planes = gpu_ctx.segment(xyz, api.default_params(k=k))
    finally:
        gpu_ctx.set_audit(False)
    tm = gpu_ctx.timings()
    assert tm["audit_mismatches"] == 0 and tm["audit_attempts"] == tm["n_seed_attempts"], tm
    print(f"[urban_200m] audit: {tm['audit_attempts']} attempts replayed in {tm['audit_ms']:.0f} ms, 0 mismatches", flush=True)
    print(f"[urban_200m] segment (host buffers) {time.time() - t0:.1f}s, device {tm['total_ms']:.0f} ms = "
          f"{len(xyz) / tm['total_ms'] / 1e3:.0f} Mpoints/s, rounds {tm['rg_rounds']}, planes {len(planes)}", flush=True)
    assert len(planes) > 5000 and int((plane_idx > 0).sum()) > 150_000_000
    check_properties(xyz, k, neigh, normals, plane_idx, planes)
