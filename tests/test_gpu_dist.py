"""Sharded single-cloud path with the HIP backend: two processes (gloo for the collectives --
one GPU cannot host two RCCL ranks -- both computing on cuda:0 through the C ABI with DEVICE
tensors) must reproduce the single-process oracle results exactly.  This is the code path of
`bench.py --gpus N` (buildingsegment_amd.dist.segment_sharded_dev); with nccl the only
difference is that `_coll` hands the device tensors to RCCL instead of staging them."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
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
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)  # torch initialises the GPU before the HIP library is loaded (as bench.py does)
    from buildingsegment_amd import api, dist as bsd, synth
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    p = api.default_params(k=16)
    xyz = synth.urban(60_000, seed=12)
    n = len(xyz)
    b = bsd.slab_bounds(n, world)
    d_xyz = torch.from_numpy(xyz[b[rank]:b[rank + 1]]).to(dev)
    d_gidx = torch.arange(b[rank], b[rank + 1], dtype=torch.int32, device=dev)
    labels, info = bsd.segment_sharded_dev(ctx, d_xyz, d_gidx, n, p, halo=250.0)
    assert labels.is_cuda and info["neigh_own"].is_cuda
    np.savez(out % rank, idx=info["gidx_own"].cpu().numpy(), ng=info["neigh_own"].cpu().numpy(),
             nr=info["normals_own"].cpu().numpy(), labels=labels.cpu().numpy(), n_local=info["n_local"],
             nplanes=-1 if info["planes"] is None else len(info["planes"]))
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
    seen = np.zeros(len(xyz), int)
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g["ng"], ng[g["idx"]])
        assert np.array_equal(g["nr"], nr[g["idx"]])
        assert np.array_equal(g["labels"], pi)
        assert g["n_local"] < len(xyz)
        assert g["nplanes"] == (len(pl["id"]) if r == 0 else -1)
        seen[g["idx"]] += 1
    assert (seen == 1).all()


def test_bench_sharded_mode_end_to_end():
    """`bench.py --gpus 2` (the driver's contract) sharding ONE cloud over two ranks: spawned by bench.py itself,
    both ranks on this box's single GPU, collectives over gloo (host-staged inside dist._coll; with nccl the
    same code hands device tensors to RCCL).  The JSON line must say what was measured."""
    import json
    import subprocess
    env = dict(os.environ, BS_CLOUD_CACHE="")
    env.pop("BS_CLOUD_CACHE")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload",
                        "plane_cube_100k", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["points_total"] == 99946 and "replicas only" in d["config"]["parallelism"]
    assert set(d["stages_ms"]) == {"partition_ms", "halo_ms", "knn_normals_ms", "gather_ms", "grow_ms"}
    # a mislabelled run is refused: WORLD_SIZE (1 rank) != --gpus
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert q.returncode == 2 and "refusing" in q.stderr
