"""bs_segment_sharded (the C ABI's multi-GPU entry point, csrc/bs_sharded.hip) on the one GPU of the test box:
  * world 2 and 3 as THREADS of this process, each with its own context and stream on cuda:0, talking through
    bs_comm_local_create (device copies behind a host barrier) -- every step of the C++ orchestration runs with
    real multi-rank data: partition, halo, certification, union-find all-reduce, deal, redistribution, growth,
    global plane ids;
  * world 1 through bs_comm_rccl with BS_SHARD_FORCE_COMM=1: the RCCL calls themselves (ncclAllReduce,
    ncclAllGather, grouped ncclSend / ncclRecv) on a communicator created from bs_comm_rccl_unique_id/_init.
Labels and plane records must equal the single-process CPU oracle bit for bit."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected(oracle, xyz, k):
    ng, nr = oracle.knn_normals(xyz, k=k)
    pi, pl = oracle.region_grow(xyz, nr, ng)
    return pi, pl


def _check_planes(all_planes, pl):
    all_planes = sorted(all_planes, key=lambda q: q.id)
    assert [q.id for q in all_planes] == pl["id"].tolist()
    off = pl["offset"]
    for i, q in enumerate(all_planes):
        assert np.array_equal(q.pointIdx, pl["point_idx"][off[i]:off[i + 1]])
        assert np.array_equal(q.center, pl["center"][i])
        assert np.array_equal(q.normal.view(np.int64), pl["normal"][i].view(np.int64))


@pytest.mark.parametrize("world", [2, 3])
def test_threads_as_ranks_on_one_gpu(oracle, gpu_ctx, world):
    import torch
    from buildingsegment_amd import _lib, api, synth
    L = _lib.load()
    xyz = synth.boxes()
    n = len(xyz)
    k = 16
    dev = torch.device("cuda", 0)
    ops = (_lib.CommOps * world)()
    assert L.bs_comm_local_create(world, ops) == 0
    bounds = [(n * r) // world for r in range(world + 1)]
    res = [None] * world

    def run(r):
        try:
            ctx = api.Context(0)
            d_xyz = torch.from_numpy(xyz[bounds[r]:bounds[r + 1]]).to(dev)
            d_g = torch.arange(bounds[r], bounds[r + 1], dtype=torch.int32, device=dev)
            d_lab = torch.empty(n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            info = ctx.segment_sharded(ops[r], d_xyz.data_ptr(), d_g.data_ptr(), len(d_xyz), n, d_lab.data_ptr(),
                                       api.default_params(k=k), halo=250.0)
            res[r] = (d_lab.cpu().numpy(), info, ctx.sharded_planes_fetch())
            ctx.close()
        except Exception as e:  # noqa: BLE001
            res[r] = e

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(world):
        L.bs_comm_local_destroy(C.byref(ops[r]))
    for r in range(world):
        assert not isinstance(res[r], Exception), res[r]
    pi, pl = _expected(oracle, xyz, k)
    planes = []
    for lab, info, pls in res:
        assert np.array_equal(lab, pi)
        assert info["components"] == 6 and info["planes_total"] == len(pl["id"])
        assert info["n_local"] < n and 0 < info["n_grow"] < n
        planes += pls
    assert sum(info["n_grow"] for _, info, _ in res) == n and sum(info["n_own"] for _, info, _ in res) == n
    _check_planes(planes, pl)


_RCCL_SNIPPET = r"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
os.environ["BS_SHARD_FORCE_COMM"] = "1"
torch.zeros(1, device="cuda")
from buildingsegment_amd import _lib, api, synth
L = _lib.load()
ctx = api.Context(0)
uid = C.create_string_buffer(128)
assert L.bs_comm_rccl_unique_id(uid) == 0
comm = C.c_void_p()
assert L.bs_comm_rccl_init(ctx._h, uid, 0, 1, C.byref(comm)) == 0
ops = _lib.CommOps()
assert L.bs_comm_rccl(comm, 0, 1, C.byref(ops)) == 0
xyz = synth.boxes()
n = len(xyz)
dev = torch.device("cuda", 0)
d_xyz = torch.from_numpy(xyz).to(dev)
d_g = torch.arange(n, dtype=torch.int32, device=dev)
d_lab = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
info = ctx.segment_sharded(ops, d_xyz.data_ptr(), d_g.data_ptr(), n, n, d_lab.data_ptr(), api.default_params(k=16))
pls = ctx.sharded_planes_fetch()
np.savez(sys.argv[2], lab=d_lab.cpu().numpy(), components=info["components"], cc=info["cc_iterations"], npl=len(pls),
         ids=np.array([q.id for q in pls]), first=np.array([q.pointIdx[0] for q in pls]))
assert L.bs_comm_rccl_destroy(comm) == 0
ctx.close()
"""


def test_rccl_communicator_at_world_one(oracle, tmp_path):
    from buildingsegment_amd import synth
    out = str(tmp_path / "rccl.npz")
    p = subprocess.run([sys.executable, "-c", _RCCL_SNIPPET, ROOT, out], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    g = np.load(out)
    xyz = synth.boxes()
    pi, pl = _expected(oracle, xyz, 16)
    assert np.array_equal(g["lab"], pi)
    assert int(g["components"]) == 6 and int(g["cc"]) >= 2  # hooked, all-reduced, then verified: two rounds
    assert g["ids"].tolist() == pl["id"].tolist()
    assert g["first"].tolist() == pl["point_idx"][pl["offset"][:-1]].tolist()
