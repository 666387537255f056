"""Synthetic integer-millimetre clouds for the buildingSegment hot path.

The reference ships no data; these generators implement the
workloads of SURVEY.md section 8(d) / BASELINE.json ``configs``.  All outputs are
``int32 [N, 3]`` millimetres, already shifted so that every coordinate is >= 0
(what ``buildingSeg``'s constructor does to the cloud before the hot path runs,
/root/reference/tmc3/TMC3.cpp:55-73).

The PRNG is counter-based splitmix64 so that C++ and Python can reproduce the
same streams: ``u64(seed, stream, i) = mix(seed + GOLDEN * (stream * 2**40 + i + 1))``.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """n 64-bit outputs of stream ``stream`` (vectorised splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(start + 1, start + n + 1, dtype=np.uint64) + np.uint64(stream) * np.uint64(1 << 40)
        z = np.uint64(seed) + _GOLDEN * ctr
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _randint(seed: int, stream: int, n: int, lo: int, hi: int) -> np.ndarray:
    """Integers uniform in [lo, hi] (inclusive); modulo bias is irrelevant here."""
    span = np.uint64(hi - lo + 1)
    return (splitmix64(seed, stream, n) % span).astype(np.int64) + lo


def _uniform(seed: int, stream: int, n: int) -> np.ndarray:
    return (splitmix64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def permutation(seed: int, stream: int, n: int) -> np.ndarray:
    """Deterministic permutation: argsort of a splitmix64 stream (stable)."""
    return np.argsort(splitmix64(seed, stream, n), kind="stable")


def _finish(pts: np.ndarray, seed: int, shuffle: bool) -> np.ndarray:
    pts = pts.astype(np.int64)
    pts -= pts.min(axis=0, keepdims=True)  # buildingSeg ctor shift (TMC3.cpp:70-72)
    if shuffle:
        pts = pts[permutation(seed, 99, len(pts))]
    assert pts.max() < (1 << 23), "synthetic cloud exceeds the exact-normal domain"
    return np.ascontiguousarray(pts.astype(np.int32))


def _face(origin, eu, ev, nu, nv, spacing, seed, stream, normal_noise=5, inplane=10):
    """nu x nv grid on the face origin + i*spacing*eu + j*spacing*ev with integer
    jitter: +-inplane along eu/ev, +-normal_noise along eu x ev."""
    eu = np.asarray(eu, dtype=np.int64)
    ev = np.asarray(ev, dtype=np.int64)
    en = np.cross(eu, ev)
    i, j = np.meshgrid(np.arange(nu, dtype=np.int64), np.arange(nv, dtype=np.int64), indexing="ij")
    i = i.ravel()
    j = j.ravel()
    m = i.size
    ju = _randint(seed, stream * 3 + 0, m, -inplane, inplane)
    jv = _randint(seed, stream * 3 + 1, m, -inplane, inplane)
    jn = _randint(seed, stream * 3 + 2, m, -normal_noise, normal_noise)
    p = (np.asarray(origin, dtype=np.int64)[None, :]
         + (i * spacing + ju)[:, None] * eu[None, :]
         + (j * spacing + jv)[:, None] * ev[None, :]
         + jn[:, None] * en[None, :])
    return p


def plane_cube(seed: int = 1, shuffle: bool = True) -> np.ndarray:
    """C0: 250x250 plane @40 mm (62 500 pts) + cube of 6 faces 79x79 @40 mm
    (37 446 pts), jitter +-5 mm on all axes; 99 946 points."""
    s = 40
    faces = [_face((0, 0, 0), (1, 0, 0), (0, 1, 0), 250, 250, s, seed, 1, 5, 5)]
    e = 78 * s  # 3.12 m
    o = np.array([3000, 3000, 1000])
    specs = [
        (o, (1, 0, 0), (0, 1, 0)), (o + (0, 0, e), (1, 0, 0), (0, 1, 0)),
        (o, (1, 0, 0), (0, 0, 1)), (o + (0, e, 0), (1, 0, 0), (0, 0, 1)),
        (o, (0, 1, 0), (0, 0, 1)), (o + (e, 0, 0), (0, 1, 0), (0, 0, 1)),
    ]
    for t, (org, eu, ev) in enumerate(specs):
        faces.append(_face(org, eu, ev, 79, 79, s, seed, 10 + t, 5, 5))
    return _finish(np.concatenate(faces), seed, shuffle)


def facade(n_side: int = 1000, seed: int = 2, shuffle: bool = True, windows: int = 10,
           recess: int = 200, spacing: int = 50) -> np.ndarray:
    """C1: n_side x n_side wall @spacing in the x-z plane; a windows x windows
    array of recesses (central half of each bay pushed back by ``recess`` mm);
    normal noise +-5 mm, in-plane jitter +-10 mm.  n_side=1000 -> exactly 1 M."""
    p = _face((0, 0, 0), (1, 0, 0), (0, 0, 1), n_side, n_side, spacing, seed, 1)
    i, j = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    bay = max(n_side // windows, 1)
    fi = (i.ravel() % bay) / bay
    fj = (j.ravel() % bay) / bay
    inside = (fi >= 0.25) & (fi < 0.75) & (fj >= 0.25) & (fj < 0.75)
    # eu x ev = (1,0,0)x(0,0,1) = (0,-1,0): push the recess along +y (into the wall)
    p[inside, 1] += recess
    return _finish(p, seed, shuffle)


def urban(n_target: int, seed: int = 3, shuffle: bool = True, spacing: int = 50,
          pitch_m: int = 40) -> np.ndarray:
    """C2/C3/C4: box buildings (footprint U[10,30] m, height U[10,40] m, four
    walls + flat roof, no ground) on a square street grid, sampled @spacing with
    the C1 noise model, truncated to exactly n_target points."""
    faces = []
    total = 0
    b = 0
    side = max(int(np.ceil(np.sqrt(max(n_target / 1.0e6, 1.0)))) + 1, 2)
    while total < n_target:
        u = _uniform(seed, 1000 + b, 3)
        ax = int(10000 + u[0] * 20000) // spacing
        ay = int(10000 + u[1] * 20000) // spacing
        hz = int(10000 + u[2] * 30000) // spacing
        ox = (b % side) * pitch_m * 1000
        oy = (b // side) * pitch_m * 1000
        o = np.array([ox, oy, 0])
        ex, ey, ez = ax * spacing, ay * spacing, hz * spacing
        specs = [
            (o, (1, 0, 0), (0, 0, 1), ax, hz), (o + (0, ey, 0), (1, 0, 0), (0, 0, 1), ax, hz),
            (o, (0, 1, 0), (0, 0, 1), ay, hz), (o + (ex, 0, 0), (0, 1, 0), (0, 0, 1), ay, hz),
            (o + (0, 0, ez), (1, 0, 0), (0, 1, 0), ax, ay),
        ]
        for t, (org, eu, ev, nu, nv) in enumerate(specs):
            f = _face(org, eu, ev, nu, nv, spacing, seed, 100 + b * 8 + t)
            faces.append(f)
            total += len(f)
        b += 1
    pts = np.concatenate(faces)[:n_target]
    return _finish(pts, seed, shuffle)


def boxes(n_boxes: int = 6, seed: int = 21, spacing: int = 50, edge_lo: int = 30, edge_hi: int = 60, pitch: int = 6000,
          shuffle: bool = True) -> np.ndarray:
    """Small multi-building scene for the sharded-path tests: n_boxes separate box "buildings" (four walls + roof,
    edges of edge_lo..edge_hi samples @spacing, C1 noise model) on a row/column grid with `pitch` mm between
    origins -- far enough apart that every box is its own connected component of the kNN graph."""
    faces = []
    side = max(int(np.ceil(np.sqrt(n_boxes))), 1)
    for b in range(n_boxes):
        u = _uniform(seed, 2000 + b, 3)
        ax, ay, hz = (int(edge_lo + u[t] * (edge_hi - edge_lo)) for t in range(3))
        o = np.array([(b % side) * pitch, (b // side) * pitch, 0])
        ex, ey, ez = ax * spacing, ay * spacing, hz * spacing
        specs = [
            (o, (1, 0, 0), (0, 0, 1), ax, hz), (o + (0, ey, 0), (1, 0, 0), (0, 0, 1), ax, hz),
            (o, (0, 1, 0), (0, 0, 1), ay, hz), (o + (ex, 0, 0), (0, 1, 0), (0, 0, 1), ay, hz),
            (o + (0, 0, ez), (1, 0, 0), (0, 1, 0), ax, ay),
        ]
        for t, (org, eu, ev, nu, nv) in enumerate(specs):
            faces.append(_face(org, eu, ev, nu, nv, spacing, seed, 300 + b * 8 + t))
    return _finish(np.concatenate(faces), seed, shuffle)


def uniform(n: int, seed: int = 6) -> np.ndarray:
    """U: n integer points i.i.d. uniform in [0, L)^3, L = round(50 * n^(1/3)) mm."""
    L = int(round(50.0 * n ** (1.0 / 3.0)))
    cols = [(splitmix64(seed, a, n) % np.uint64(L)).astype(np.int32) for a in range(3)]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def shift_to_origin(xyz: np.ndarray) -> np.ndarray:
    """Mirror of the buildingSeg constructor's shift (TMC3.cpp:55-73)."""
    xyz = np.asarray(xyz)
    return np.ascontiguousarray((xyz.astype(np.int64) - xyz.min(axis=0, keepdims=True)).astype(np.int32))


WORKLOADS = {
    "plane_cube_100k": lambda: plane_cube(),
    "facade_1m": lambda: facade(),
    "urban_10m": lambda: urban(10_000_000, seed=3),
    "urban_50m": lambda: urban(50_000_000, seed=4),
    "uniform_1m": lambda: uniform(1_000_000),
}
