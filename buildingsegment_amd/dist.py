"""Single-node multi-GPU layer: ONE cloud sharded over the ranks (one process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests and in one-GPU rehearsals).

What shards and what does not (reference loops:
/root/reference/tmc3/my_function.h:63 normals, :71-78 kNN; the ordered scan
that stays serial: /root/reference/tmc3/my_function.cpp:184-217):

  * stages 1-2 (kNN + normals) are independent per query after ONE exchange step.
    Every rank starts from an arbitrary 1/N of the cloud (+ global indices):
      1. Morton partition: 63-bit keys on the device, splitters from gathered
         samples, ONE all-to-all that moves every point to its slab owner;
      2. halo: each rank publishes the voxels (edge >= h) its slab occupies; a
         point is sent to peer r iff its voxel touches (27-neighbourhood) a voxel
         of r -- that covers every point within h of any point of r for ANY slab
         shape; ONE all-to-all of (x, y, z, gidx) rows;
      3. slab kNN + normals (bs_knn_normals_dev with d_gidx: ties by GLOBAL index,
         global indices out).  A k-list is exact iff its k-th distance is < h;
         ranks agree (all-reduce MAX) to retry with a doubled halo otherwise.
  * stage 3 (region growing) is order dependent and sequential per plane: it is
    NOT sharded -- "replicas only".  xyz / neigh / normals are sent to rank 0
    (one all-to-all whose only receiver is rank 0), rank 0 grows, labels are
    broadcast.

Every buffer stays on the compute device: with nccl the collectives run on the
device tensors themselves (no .cpu(), no numpy); with gloo the SAME code stages
each collective through host memory inside `_coll` (gloo has no device
transport) -- that is the only difference between the two backends.

The compute backend is injected so that the orchestration is covered on CPU:
    knn_normals(xyz_loc [n,3] i32, gidx_loc [n] i32, n_query, params, cert_radius)
        -> (neigh [n_query,k] i32 GLOBAL indices, normals [n_query,3] f64, n_uncertified)
    region_grow(xyz [n,3], normals [n,3], neigh [n,k], params) -> (labels [n] i32, planes | None)
`DevBackend` wraps an api.Context (HIP, device pointers); tests/test_dist_gloo.py
injects a CPU backend of its own.
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch
import torch.distributed as dist

_OFFS27 = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]


# --------------------------------------------------------------------------
# Morton order (host helper kept for tests / tools; the device path uses
# morton_keys_t on tensors)
# --------------------------------------------------------------------------

def _spread_t(v: torch.Tensor) -> torch.Tensor:
    v = v & 0x1FFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def morton_keys_t(xyz: torch.Tensor, origin: torch.Tensor, shift: int) -> torch.Tensor:
    """63-bit Morton keys (int64) of int32 coordinates: 21 bits per axis after the
    shift to `origin` and dropping `shift` low bits."""
    x = (xyz.to(torch.int64) - origin.to(torch.int64)[None, :]) >> shift
    return _spread_t(x[:, 0]) | (_spread_t(x[:, 1]) << 1) | (_spread_t(x[:, 2]) << 2)


def morton_keys(xyz: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.int32))
    origin = t.min(0).values.to(torch.int64)
    ext = int((t.to(torch.int64) - origin[None, :]).max()) if len(t) else 0
    return morton_keys_t(t, origin, max(ext.bit_length() - 21, 0)).numpy()


def slab_bounds(n: int, world: int) -> list[int]:
    """Equal-count contiguous ranges."""
    return [(n * r) // world for r in range(world + 1)]


def partition_morton(xyz: np.ndarray, world: int, rank: int):
    """Slab r of the Morton order of a replicated cloud: (xyz_own, gidx_own)."""
    order = np.argsort(morton_keys(xyz), kind="stable")
    b = slab_bounds(len(xyz), world)
    own = np.sort(order[b[rank]:b[rank + 1]])
    return np.ascontiguousarray(xyz[own]), own.astype(np.int32)


# --------------------------------------------------------------------------
# collectives on device tensors
# --------------------------------------------------------------------------

def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _host_staged(group=None) -> bool:
    return _world(group) > 1 and dist.get_backend(group) != "nccl"


def _coll(t: torch.Tensor, group=None) -> torch.Tensor:
    """The tensor a collective runs on: the device tensor itself with nccl; a host copy with gloo."""
    return t.cpu() if (_host_staged(group) and t.is_cuda) else t


def all_reduce_(t: torch.Tensor, op, group=None) -> torch.Tensor:
    if _world(group) == 1:
        return t
    c = _coll(t, group)
    dist.all_reduce(c, op=op, group=group)
    if c is not t:
        t.copy_(c)
    return t


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of equally shaped tensors, concatenated along dim 0."""
    w = _world(group)
    if w == 1:
        return t
    c = _coll(t.contiguous(), group)
    out = torch.empty((w * c.shape[0],) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
    dist.all_gather_into_tensor(out, c, group=group) if c.is_cuda else dist.all_gather(list(out.chunk(w)), c, group=group)
    return out.to(t.device)


def all_to_all_rows(rows: torch.Tensor, send_counts: torch.Tensor, group=None):
    """rows [m, c] sorted by destination rank, send_counts [world] (int64, on rows.device).
    Returns (received rows, recv_counts list).  Counts travel first (tiny all-to-all), then ONE
    payload all-to-all with split sizes -- only what a peer needs crosses xGMI."""
    w = _world(group)
    if w == 1:
        return rows, [int(rows.shape[0])]
    sc = _coll(send_counts.to(torch.int64), group)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    s_list = [int(v) for v in sc.tolist()]
    r_list = [int(v) for v in rc.tolist()]
    src = _coll(rows.contiguous(), group)
    out = torch.empty((sum(r_list),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=src.device)
    dist.all_to_all_single(out, src, output_split_sizes=r_list, input_split_sizes=s_list, group=group)
    return out.to(rows.device), r_list


def _sync(t: torch.Tensor):
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


# --------------------------------------------------------------------------
# compute backend over the C ABI (device pointers)
# --------------------------------------------------------------------------

class DevBackend:
    """api.Context as the slab compute backend: tensors live on the context's GPU and go
    through bs_knn_normals_dev / bs_region_grow_dev as raw device pointers."""

    def __init__(self, ctx):
        self.ctx = ctx

    def knn_normals(self, xyz_loc, gidx_loc, n_query, params, cert_radius):
        n = int(xyz_loc.shape[0])
        neigh = torch.empty((n_query, params.k), dtype=torch.int32, device=xyz_loc.device)
        normals = torch.empty((n_query, 3), dtype=torch.float64, device=xyz_loc.device)
        _sync(xyz_loc)  # inputs were produced on torch's stream; the context may use its own
        unc = self.ctx.knn_normals_dev(xyz_loc.data_ptr(), n, neigh.data_ptr(), normals.data_ptr(), params,
                                       q_begin=0, q_end=n_query, d_gidx=gidx_loc.data_ptr(), cert_radius=cert_radius)
        return neigh, normals, int(unc)

    def region_grow(self, xyz, normals, neigh, params):
        n = int(xyz.shape[0])
        labels = torch.empty((n,), dtype=torch.int32, device=xyz.device)
        _sync(xyz)
        self.ctx.region_grow_dev(xyz.data_ptr(), normals.data_ptr(), neigh.data_ptr(), n, labels.data_ptr(), params)
        return labels, self.ctx.planes_fetch()


# --------------------------------------------------------------------------
# the sharded pass
# --------------------------------------------------------------------------

def _partition(rows: torch.Tensor, world: int, group=None, samples: int = 1024):
    """rows [m,4] = x,y,z,gidx (int32).  Moves every point to its Morton-slab owner."""
    dev = rows.device
    big = torch.iinfo(torch.int64).max
    mn = rows[:, :3].min(0).values.to(torch.int64) if rows.shape[0] else torch.full((3,), 1 << 40, dtype=torch.int64, device=dev)
    mx = rows[:, :3].max(0).values.to(torch.int64) if rows.shape[0] else torch.full((3,), -(1 << 40), dtype=torch.int64, device=dev)
    all_reduce_(mn, dist.ReduceOp.MIN, group)
    all_reduce_(mx, dist.ReduceOp.MAX, group)
    ext = int((mx - mn).max().item())
    shift = max(ext.bit_length() - 21, 0)
    if world == 1:
        return rows, mn
    keys = morton_keys_t(rows[:, :3], mn, shift)
    skeys, order = torch.sort(keys)
    m = int(rows.shape[0])
    # splitters: every rank contributes `samples` evenly spaced keys of its sorted local order
    if m:
        pos = torch.linspace(0, m - 1, samples, device=dev).round().to(torch.int64)
        samp = skeys[pos]
    else:
        samp = torch.full((samples,), big, dtype=torch.int64, device=dev)
    allsamp = torch.sort(all_gather_rows(samp, group)).values
    valid = int((allsamp != big).sum().item())
    cut = [min(valid - 1, max(0, (valid * r) // world)) for r in range(1, world)]
    splitters = allsamp[torch.tensor(cut, dtype=torch.int64, device=dev)] if valid else allsamp[:world - 1]
    dest = torch.bucketize(skeys, splitters, right=True)  # sorted keys -> non-decreasing destinations
    counts = torch.bincount(dest, minlength=world)
    got, _ = all_to_all_rows(rows[order], counts, group)
    return got, mn


def _halo(own: torch.Tensor, origin: torch.Tensor, h: float, world: int, rank: int, group=None):
    """own [m,4].  Returns the halo rows [*,4] this rank receives for halo width h."""
    dev = own.device
    if world == 1:
        return own[:0]
    v = max(int(math.ceil(h)), 500)
    vox = torch.div(own[:, :3].to(torch.int64) - origin[None, :], v, rounding_mode="floor") + 1  # +1: room for the -1 offsets
    vkey = (vox[:, 0] << 42) | (vox[:, 1] << 21) | vox[:, 2]
    occ = torch.unique(vkey)
    cnt = torch.tensor([occ.shape[0]], dtype=torch.int64, device=dev)
    cnts = all_gather_rows(cnt, group).tolist()
    cap = max(max(cnts), 1)
    pad = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    pad[:occ.shape[0]] = occ
    allocc = all_gather_rows(pad, group).reshape(world, cap)
    offs = torch.tensor([(dx << 42) + (dy << 21) + dz for dx, dy, dz in _OFFS27], dtype=torch.int64, device=dev)
    parts, counts = [], []
    for r in range(world):
        if r == rank or own.shape[0] == 0 or cnts[r] == 0:
            counts.append(0)
            continue
        dil = torch.unique((allocc[r, :cnts[r], None] + offs[None, :]).reshape(-1))
        sel = torch.isin(vkey, dil)
        parts.append(own[sel])
        counts.append(int(sel.sum().item()))
    send = torch.cat(parts) if parts else own[:0]
    got, _ = all_to_all_rows(send, torch.tensor(counts, dtype=torch.int64, device=dev), group)
    return got  # a point has one owner and is sent to a peer at most once: no duplicates


def segment_sharded_dev(backend, d_xyz: torch.Tensor, d_gidx: torch.Tensor, n_total: int, params, halo: float = 0.0,
                        group=None, max_retries: int = 6, grow: bool = True):
    """Segment ONE cloud whose points are spread over the ranks (any split; rank r passes its
    points `d_xyz` int32 [m,3] and their global indices `d_gidx` int32 [m], resident on its device).

    Returns (labels, info): labels = int32 [n_total] tensor, identical on every rank (None if
    grow=False); info carries the slab results (`gidx_own`, `neigh_own`, `normals_own`: this
    rank's Morton slab), `planes` (rank 0 only -- stage 3 is "replicas only", the other ranks
    get None), the halo width used, the retries and per-stage wall times."""
    if hasattr(backend, "knn_normals_dev"):  # an api.Context: wrap it
        backend = DevBackend(backend)
    world, rank = _world(group), _rank(group)
    dev = d_xyz.device
    k = params.k
    st = {}

    def tick(name, t0):
        _sync(d_xyz)
        st[name] = st.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    t0 = time.perf_counter()
    rows = torch.cat([d_xyz.to(torch.int32), d_gidx.to(torch.int32)[:, None]], 1)
    own, origin = _partition(rows, world, group)
    n_own = int(own.shape[0])
    t0 = tick("partition_ms", t0)
    h = float(halo) if halo > 0 else 2.0 * float(params.radius)
    h = max(h, float(params.radius))  # the hybrid radius must be covered
    retries = 0
    st["halo_ms"] = st["knn_normals_ms"] = 0.0
    while True:
        hal = _halo(own, origin, h, world, rank, group)
        loc = torch.cat([own, hal]) if hal.shape[0] else own
        xyz_loc = loc[:, :3].contiguous()
        gidx_loc = loc[:, 3].contiguous()
        t0 = tick("halo_ms", t0)
        if n_own and xyz_loc.shape[0] >= k:
            neigh, normals, unc = backend.knn_normals(xyz_loc, gidx_loc, n_own, params, h if world > 1 else 0.0)
        else:  # a slab that cannot even fill one k-list: force a wider halo (or fail below)
            neigh = torch.zeros((n_own, k), dtype=torch.int32, device=dev)
            normals = torch.zeros((n_own, 3), dtype=torch.float64, device=dev)
            unc = n_own
        t0 = tick("knn_normals_ms", t0)
        flag = torch.tensor([unc], dtype=torch.int64, device=dev)
        all_reduce_(flag, dist.ReduceOp.MAX, group)
        if int(flag.item()) == 0:
            break
        retries += 1
        if retries > max_retries or world == 1:
            raise RuntimeError("halo exchange could not certify every k-list "
                               f"(halo {h} mm after {retries - 1} doublings; n_total={n_total}, k={k})")
        h *= 2.0  # a thin halo is a performance matter, never a correctness one
    info = {"halo": h, "retries": retries, "n_local": int(xyz_loc.shape[0]), "n_own": n_own,
            "gidx_own": own[:, 3].contiguous(), "neigh_own": neigh, "normals_own": normals, "planes": None}
    labels = None
    st["gather_ms"] = st["grow_ms"] = 0.0
    if grow:
        # stage 3, replicas only: the graph goes to rank 0 (the only receiver), which grows and broadcasts
        if world > 1:
            to0 = torch.zeros(world, dtype=torch.int64, device=dev)
            to0[0] = n_own
            g_own, _ = all_to_all_rows(own, to0, group)
            g_ng, _ = all_to_all_rows(neigh, to0, group)
            g_nr, _ = all_to_all_rows(normals, to0, group)
        else:
            g_own, g_ng, g_nr = own, neigh, normals
        labels = torch.empty(n_total, dtype=torch.int32, device=dev)
        if rank == 0:
            gi = g_own[:, 3].to(torch.int64)
            if g_own.shape[0] != n_total or int(torch.unique(gi).shape[0]) != n_total:
                raise RuntimeError("partition must cover the cloud exactly once")
            xyz_all = torch.empty((n_total, 3), dtype=torch.int32, device=dev)
            neigh_all = torch.empty((n_total, k), dtype=torch.int32, device=dev)
            normals_all = torch.empty((n_total, 3), dtype=torch.float64, device=dev)
            xyz_all[gi] = g_own[:, :3]
            neigh_all[gi] = g_ng
            normals_all[gi] = g_nr
            del g_own, g_ng, g_nr
            t0 = tick("gather_ms", t0)
            lab, planes = backend.region_grow(xyz_all, normals_all, neigh_all, params)
            labels.copy_(lab)
            info["planes"] = planes
            t0 = tick("grow_ms", t0)
        if world > 1:
            c = _coll(labels, group)
            dist.broadcast(c, src=0, group=group)
            if c is not labels:
                labels.copy_(c)
        t0 = tick("gather_ms" if rank else "grow_ms", t0)
    info["stage_ms"] = st
    return labels, info
