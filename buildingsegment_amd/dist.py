"""Single-node multi-GPU layer (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

What shards and what does not:
  * stages 1-2 (kNN + normals) are independent per query after ONE exchange
    step: the cloud is cut into equal-count slabs of its Morton order, every
    rank receives the points inside its slab's bounding box grown by the halo
    width h (all-gather of padded halo buffers), and runs the slab form of the
    kernel (bs_knn_normals_halo).  A k-list is exact iff its k-th distance is
    < h; ranks agree (all-reduce MAX) to retry with a doubled halo otherwise.
  * stage 3 (region growing) is order dependent and sequential per plane: it is
    NOT sharded ("replicas only").  neigh + normals are all-gathered into
    global index order, rank 0 grows, labels are broadcast.

The compute backend is injected so that the N>1 path can be covered on CPU:
any object with
    knn_normals_halo(xyz_local, gidx, n_query, params, cert_radius) -> (neigh, normals, n_uncertified)
    region_grow(xyz, normals, neigh, params) -> (plane_idx, planes)
(buildingsegment_amd.api.Context is one).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def morton_keys(xyz: np.ndarray) -> np.ndarray:
    """63-bit Morton keys of int32 coordinates (shifted to >= 0, 21 bits per axis
    after dropping low bits when the extent needs more)."""
    x = xyz.astype(np.int64) - xyz.min(axis=0, keepdims=True).astype(np.int64)
    shift = max(int(x.max()).bit_length() - 21, 0)
    x >>= shift

    def spread(v):
        v = v & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v

    return spread(x[:, 0]) | (spread(x[:, 1]) << 1) | (spread(x[:, 2]) << 2)


def slab_bounds(n: int, world: int) -> list[int]:
    """Equal-count contiguous ranges of the sorted order."""
    return [(n * r) // world for r in range(world + 1)]


def _dev(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def all_gather_varlen(t: torch.Tensor, group=None) -> list[torch.Tensor]:
    """all-gather of tensors whose first dimension differs per rank: counts first,
    then one padded all-gather (fewer, larger collectives suit xGMI)."""
    world = dist.get_world_size(group)
    dev = _dev(group)
    t = t.to(dev).contiguous()
    cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c.item()) for c in cnts]
    m = max(max(cnts), 1)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:c].cpu() for o, c in zip(out, cnts)]


def segment_sharded(xyz_own: np.ndarray, gidx_own: np.ndarray, n_total: int, backend, params, halo: float = 0.0,
                    group=None, max_retries: int = 6):
    """Segment ONE cloud whose points are partitioned over the ranks.

    xyz_own / gidx_own: this rank's slab (any partition works; slabs of the
    Morton order keep halos small) and the global indices of its points.
    Returns (neigh_own, normals_own, plane_idx_global, planes, info) -- labels
    and planes are identical on every rank.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = _dev(group)
    xyz_own = np.ascontiguousarray(xyz_own, dtype=np.int32)
    gidx_own = np.ascontiguousarray(gidx_own, dtype=np.int32)
    n_own = len(xyz_own)
    h = float(halo) if halo > 0 else 2.0 * float(params.radius)
    h = max(h, float(params.radius))  # the hybrid radius must be covered
    # global origin so that every rank voxelises identically
    mn = torch.from_numpy(xyz_own.min(0).astype(np.int64) if n_own else np.full(3, 1 << 40, np.int64)).to(dev)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    origin = mn.cpu().numpy()
    retries = 0
    while True:
        # Each rank publishes the voxels (edge v >= h) its slab occupies; a point is
        # sent to peer r iff its voxel touches (27-neighbourhood) one of r's voxels.
        # That covers every point within h of any point of r for ANY slab shape
        # (Morton slabs are not boxes).
        v = max(int(np.ceil(h)), 500)
        vox = ((xyz_own.astype(np.int64) - origin) // v) + 1  # +1: room for the -1 offsets
        vkey = (vox[:, 0] << 42) | (vox[:, 1] << 21) | vox[:, 2]
        occ = all_gather_varlen(torch.from_numpy(np.unique(vkey)), group)
        offs = np.array([(dx << 42) + (dy << 21) + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)],
                        np.int64)
        rows = []
        for r in range(world):
            if r == rank or n_own == 0 or len(occ[r]) == 0:
                continue
            dil = np.unique((occ[r].numpy()[:, None] + offs[None, :]).ravel())
            m = np.isin(vkey, dil)
            if m.any():
                rows.append(np.concatenate([xyz_own[m], gidx_own[m, None], np.full((m.sum(), 1), r, np.int32)], 1))
        send = np.concatenate(rows) if rows else np.zeros((0, 5), np.int32)
        got = all_gather_varlen(torch.from_numpy(send.astype(np.int32)), group)
        halo_rows = [g.numpy() for r, g in enumerate(got) if r != rank]
        halo_pts = np.concatenate(halo_rows) if halo_rows else np.zeros((0, 5), np.int32)
        halo_pts = halo_pts[halo_pts[:, 4] == rank]
        # the same point can never arrive twice (each point has one owner)
        xyz_loc = np.concatenate([xyz_own, halo_pts[:, :3]]).astype(np.int32)
        gidx_loc = np.concatenate([gidx_own, halo_pts[:, 3]]).astype(np.int32)
        if len(xyz_loc) >= params.k and n_own:
            neigh, normals, unc = backend.knn_normals_halo(xyz_loc, gidx_loc, n_own, params, h)
        elif n_own:
            neigh, normals, unc = np.zeros((n_own, params.k), np.int32), np.zeros((n_own, 3)), n_own
        else:
            neigh, normals, unc = np.zeros((0, params.k), np.int32), np.zeros((0, 3)), 0
        flag = torch.tensor([unc], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            break
        retries += 1
        if retries > max_retries:
            raise RuntimeError("halo exchange could not certify every k-list")
        h *= 2.0  # a thin halo is a performance matter, never a correctness one
    # stage 3: whole graph on every rank in global index order, rank 0 grows, labels broadcast
    g_idx = np.concatenate([g.numpy() for g in all_gather_varlen(torch.from_numpy(gidx_own), group)])
    g_xyz = np.concatenate([g.numpy() for g in all_gather_varlen(torch.from_numpy(xyz_own), group)])
    g_ng = np.concatenate([g.numpy() for g in all_gather_varlen(torch.from_numpy(neigh), group)])
    g_nr = np.concatenate([g.numpy() for g in all_gather_varlen(torch.from_numpy(normals), group)])
    assert len(g_idx) == n_total and len(np.unique(g_idx)) == n_total, "partition must cover the cloud exactly once"
    inv = np.empty(n_total, np.int64)
    inv[g_idx] = np.arange(n_total)
    xyz_all, neigh_all, normals_all = g_xyz[inv], g_ng[inv], g_nr[inv]
    labels = torch.empty(n_total, dtype=torch.int32, device=dev)
    planes = None
    if rank == 0:
        plane_idx, planes = backend.region_grow(xyz_all, normals_all, neigh_all, params)
        labels.copy_(torch.from_numpy(np.ascontiguousarray(plane_idx, dtype=np.int32)))
    dist.broadcast(labels, src=0, group=group)
    info = {"halo": h, "retries": retries, "n_local": len(xyz_loc), "n_own": n_own}
    return neigh, normals, labels.cpu().numpy(), planes, info


def partition_morton(xyz: np.ndarray, world: int, rank: int):
    """Slab r of the Morton order of a replicated cloud: (xyz_own, gidx_own)."""
    order = np.argsort(morton_keys(xyz), kind="stable")
    b = slab_bounds(len(xyz), world)
    own = np.sort(order[b[rank]:b[rank + 1]])
    return np.ascontiguousarray(xyz[own]), own.astype(np.int32)
