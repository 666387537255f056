"""N>1 path on CPU: world_size-2/3 gloo runs of buildingsegment_amd.dist.segment_sharded_dev
with the CPU oracle injected as compute backend (the HIP backend needs a GPU).  The
orchestration under test is the SAME code bench.py --gpus N runs on device tensors: Morton
partition (all-to-all), voxel halo (all-to-all), certification + retry, graph to rank 0,
rank-0 region grow, label broadcast."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Slab semantics of bs_knn_normals_dev(d_gidx, cert_radius) restated with the CPU oracle:
    ties by GLOBAL index (local cloud sorted by gidx first), global indices out,
    certification = k-th distance < cert_radius."""

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def knn_normals(self, xyz_loc, gidx_loc, n_query, params, cert_radius):
        xyz_local, gidx = xyz_loc.numpy(), gidx_loc.numpy()
        order = np.argsort(gidx, kind="stable")
        rank_of = np.empty(len(order), np.int64)
        rank_of[order] = np.arange(len(order))
        xs, gs = np.ascontiguousarray(xyz_local[order]), gidx[order]
        ng, nr = self.O.knn_normals(xs, k=params.k, radius=params.radius, max_nn=params.max_nn)
        q = rank_of[:n_query]
        ngq, nrq = ng[q], nr[q]
        unc = 0
        if cert_radius > 0:
            d = xs[ngq[:, -1]].astype(np.int64) - xs[q].astype(np.int64)
            unc = int(((d * d).sum(1) >= cert_radius * cert_radius).sum())
        return torch.from_numpy(gs[ngq].astype(np.int32)), torch.from_numpy(nrq), unc

    def region_grow(self, xyz, normals, neigh, params):
        pi, pl = self.O.region_grow(xyz.numpy(), normals.numpy(), neigh.numpy(), th_thickness=params.th_thickness,
                                    th_point_count=params.th_point_count, cos_th=params.cos_th)
        return torch.from_numpy(pi), pl


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(k=15):
    from buildingsegment_amd import _lib
    return _lib.Params(k=k, max_nn=50, radius=100.0, th_thickness=300, th_point_count=400, cos_th=0.88,
                       cell_size=0, rg_mode=0)


def _worker(rank, world, port, halo, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from buildingsegment_amd import dist as bsd, synth
    xyz = synth.plane_cube()[:24000].copy()
    n = len(xyz)
    b = bsd.slab_bounds(n, world)  # the input split is arbitrary: a contiguous 1/N of the input order
    d_xyz = torch.from_numpy(xyz[b[rank]:b[rank + 1]])
    d_gidx = torch.arange(b[rank], b[rank + 1], dtype=torch.int32)
    labels, info = bsd.segment_sharded_dev(OracleBackend(), d_xyz, d_gidx, n, _params(), halo=halo)
    np.savez(out % rank, idx=info["gidx_own"].numpy(), ng=info["neigh_own"].numpy(), nr=info["normals_own"].numpy(),
             labels=labels.numpy(), retries=info["retries"], n_local=info["n_local"],
             nplanes=-1 if info["planes"] is None else len(info["planes"]["id"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, 250.0), (2, 30.0), (3, 200.0), (5, 150.0)])
def test_sharded_run_equals_single_process(oracle, tmp_path, world, halo):
    from buildingsegment_amd import synth
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), halo, out), nprocs=world, join=True)
    xyz = synth.plane_cube()[:24000].copy()
    ng, nr = oracle.knn_normals(xyz, k=15)
    pi, pl = oracle.region_grow(xyz, nr, ng)
    seen = np.zeros(len(xyz), int)
    sizes = []
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g["ng"], ng[g["idx"]])      # neighbour indices bit-exact
        assert np.array_equal(g["nr"], nr[g["idx"]])      # normals bit-exact
        assert np.array_equal(g["labels"], pi)            # labels identical on every rank
        assert g["n_local"] < len(xyz)                    # a slab + halo, not the whole cloud
        assert g["nplanes"] == (len(pl["id"]) if r == 0 else -1)  # stage 3: replicas only, planes on rank 0
        seen[g["idx"]] += 1
        sizes.append(len(g["idx"]))
        if halo < 100:
            assert g["retries"] >= 1                      # halo below the hybrid radius is widened
    assert (seen == 1).all()                              # the Morton slabs partition the cloud
    assert max(sizes) < 1.25 * len(xyz) / world           # ... into nearly equal counts


def test_single_process_path_without_process_group(oracle):
    """world 1 (no process group): the same function degenerates to the plain pipeline."""
    from buildingsegment_amd import dist as bsd, synth
    xyz = synth.plane_cube()[:9000].copy()
    labels, info = bsd.segment_sharded_dev(OracleBackend(), torch.from_numpy(xyz), torch.arange(len(xyz), dtype=torch.int32),
                                           len(xyz), _params())
    ng, nr = oracle.knn_normals(xyz, k=15)
    pi, _ = oracle.region_grow(xyz, nr, ng)
    assert np.array_equal(info["neigh_own"].numpy(), ng) and np.array_equal(labels.numpy(), pi)


def test_morton_partition_is_a_partition():
    from buildingsegment_amd import dist as bsd, synth
    xyz = synth.uniform(5000, seed=3)
    parts = [bsd.partition_morton(xyz, 4, r)[1] for r in range(4)]
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(5000))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    k = bsd.morton_keys(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], np.int32))
    assert k.tolist() == [0, 1, 2, 4, 7]
