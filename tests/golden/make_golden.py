#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  BUILD CONTAINER ONLY: it runs
oracle/_ref/ref_stage3, the reference's own seg_plane code compiled verbatim
from /root/reference/tmc3/my_function.{h,cpp} (oracle/ref/build_ref.sh), on
seeded inputs and stores inputs + the reference's outputs.  The fixtures are
data (inputs and expected outputs); no reference source text is stored.

Stage 1-2 inputs (neigh, normals) come from the CPU oracle or from closed-form
constructions -- the reference has no runnable implementation of those stages
here (Open3D absent), so they are just *inputs* to the pinned stage 3.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from buildingsegment_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, xyz, normals, neigh):
    pi, pl, col = O.ref_region_grow(xyz, normals, neigh)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), xyz=xyz.astype(np.int32), normals=normals,
                        neigh=neigh.astype(np.int32), plane_idx=pi, id=pl["id"], normal=pl["normal"],
                        center=pl["center"], offset=pl["offset"], point_idx=pl["point_idx"], colors=col)
    print(name, "n", len(xyz), "planes", len(pl["id"]), "sizes", np.diff(pl["offset"])[:8],
          "labelled", (pi > 0).sum())


def grid_patch():
    """SURVEY probe P1: 40x40 grid @50 mm (normals +z) + 10x10 perpendicular patch."""
    g = np.stack(np.meshgrid(np.arange(40) * 50, np.arange(40) * 50, indexing="ij"), -1).reshape(-1, 2)
    a = np.concatenate([g, np.zeros((1600, 1), np.int64)], 1)
    p = np.stack(np.meshgrid(np.arange(10) * 50, np.arange(10) * 50, indexing="ij"), -1).reshape(-1, 2)
    b = np.stack([np.full(100, 5000), p[:, 0], p[:, 1] + 50], 1)
    xyz = np.concatenate([a, b]).astype(np.int32)
    normals = np.zeros((1700, 3))
    normals[:1600, 2] = 1.0
    normals[1600:, 0] = 1.0
    neigh = O.knn_brute(xyz, k=15)
    return xyz, normals, neigh


def walls(n_per, seed, offset=0, noise=0.25):
    """SURVEY probe P6: three perpendicular noisy walls, noisy normals."""
    rng = np.random.default_rng(seed)
    m = int(np.sqrt(n_per))
    u, v = np.meshgrid(np.arange(m) * 40, np.arange(m) * 40, indexing="ij")
    u = u.ravel() + rng.integers(-10, 11, m * m)
    v = v.ravel() + rng.integers(-10, 11, m * m)
    w = rng.integers(-40, 41, m * m)
    faces = [np.stack([u, v, w], 1), np.stack([u, w, v + 100], 1), np.stack([w, u + 100, v + 100], 1)]
    tn = [np.array([0, 0, 1.0]), np.array([0, 1.0, 0]), np.array([1.0, 0, 0])]
    xyz = np.concatenate(faces).astype(np.int64)
    nrm = np.concatenate([np.tile(t, (m * m, 1)) for t in tn]) + rng.normal(0, noise, (3 * m * m, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[nrm[:, 2] < 0] *= -1
    perm = rng.permutation(len(xyz))
    xyz, nrm = xyz[perm], nrm[perm]
    xyz -= xyz.min(0)
    xyz[:, 0] += offset
    xyz = xyz.astype(np.int32)
    neigh = O.knn_normals(xyz, k=15, want_normals=False)[0]
    return xyz, np.ascontiguousarray(nrm), neigh


def main():
    if O.ref_stage3_path() is None:
        sys.exit("oracle/_ref/ref_stage3 missing: run `make -C oracle ref` in the build container")
    save("grid_patch_p1", *grid_patch())
    save("walls_6k", *walls(2000, 7))
    save("walls_6k_overflow", *walls(2000, 8, offset=3_000_000))
    save("walls_9k_offset", *walls(3000, 9, offset=900_000, noise=0.1))
    # orphan probe P5: clean jittered plane, one neighbour of seed 0 gets a foreign normal
    xyz, nrm, neigh = walls(2500, 10, noise=0.0)
    keep = np.arange(len(xyz))[np.abs(nrm[:, 2]) > 0.9][:2500]
    xyz, nrm = np.ascontiguousarray(xyz[keep]), np.ascontiguousarray(nrm[keep])
    neigh = O.knn_brute(xyz, k=15)
    nrm[neigh[0, 3]] = (1.0, 0.0, 0.0)
    save("orphans_p5", xyz, nrm, neigh)
    # full pipeline inputs from the oracle's own stage 1-2 on a C0 subsample (k=15 reference literals)
    xyz = synth.plane_cube()[:12000].copy()
    neigh, nrm = O.knn_normals(xyz, k=15)
    save("plane_cube_12k", xyz, nrm, neigh)
    # k=16 facade crop
    xyz = synth.facade(n_side=100, seed=2)
    neigh, nrm = O.knn_normals(xyz, k=16)
    save("facade_10k_k16", xyz, nrm, neigh)


if __name__ == "__main__":
    main()
