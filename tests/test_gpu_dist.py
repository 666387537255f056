"""Sharded single-cloud path with the HIP backend: two processes (gloo for the
collectives, both computing on cuda:0 through the C ABI) must reproduce the
single-process oracle results exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from buildingsegment_amd import api, dist as bsd, synth
    ctx = api.Context(0)
    p = api.default_params(k=16)
    xyz = synth.urban(60_000, seed=12)
    own_xyz, own_idx = bsd.partition_morton(xyz, world, rank)
    ng, nr, labels, planes, info = bsd.segment_sharded(own_xyz, own_idx, len(xyz), ctx, p, halo=250.0)
    np.savez(out % rank, idx=own_idx, ng=ng, nr=nr, labels=labels, n_local=info["n_local"],
             nplanes=-1 if planes is None else len(planes))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_two_process_sharded_run_on_one_gpu(oracle, tmp_path):
    from buildingsegment_amd import synth
    world = 2
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    xyz = synth.urban(60_000, seed=12)
    ng, nr = oracle.knn_normals(xyz, k=16)
    pi, pl = oracle.region_grow(xyz, nr, ng)
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g["ng"], ng[g["idx"]])
        assert np.array_equal(g["nr"], nr[g["idx"]])
        assert np.array_equal(g["labels"], pi)
        assert g["n_local"] < len(xyz)
        if r == 0:
            assert g["nplanes"] == len(pl["id"])
