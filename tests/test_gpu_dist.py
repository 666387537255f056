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


def _worker(rank, world, port, out, backend_name):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    if backend_name == "nccl":  # RCCL itself, at the only world size one GPU can host: every collective of the path is issued
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.zeros(1, device=dev)  # torch initialises the GPU before the HIP library is loaded (as bench.py does)
    from buildingsegment_amd import api, dist as bsd, synth
    bsd.FORCE_COLLECTIVES = True
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    p = api.default_params(k=16)
    xyz = synth.boxes()
    n = len(xyz)
    b = bsd.slab_bounds(n, world)
    d_xyz = torch.from_numpy(xyz[b[rank]:b[rank + 1]]).to(dev)
    d_gidx = torch.arange(b[rank], b[rank + 1], dtype=torch.int32, device=dev)
    labels, info = bsd.segment_sharded_dev(ctx, d_xyz, d_gidx, n, p, halo=250.0, want_planes=True)
    assert labels.is_cuda and info["neigh_own"].is_cuda
    planes = bsd.gather_planes(info["planes"])
    np.savez(out % rank, idx=info["gidx_own"].cpu().numpy(), ng=info["neigh_own"].cpu().numpy(),
             nr=info["normals_own"].cpu().numpy(), labels=labels.cpu().numpy(), n_local=info["n_local"],
             n_grow=info["n_grow"], components=-1 if info["components"] is None else info["components"],
             pid=np.array([q["id"] for q in planes], np.int32),
             pnormal=np.array([q["normal"] for q in planes], np.float64).reshape(-1, 3),
             pcenter=np.array([q["center"] for q in planes], np.int32).reshape(-1, 3),
             poff=np.cumsum([0] + [len(q["pointIdx"]) for q in planes]),
             pidx=np.concatenate([q["pointIdx"] for q in planes]) if planes else np.zeros(0, np.int32))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def _check(oracle, out, world):
    from buildingsegment_amd import synth
    xyz = synth.boxes()
    ng, nr = oracle.knn_normals(xyz, k=16)
    pi, pl = oracle.region_grow(xyz, nr, ng)
    seen = np.zeros(len(xyz), int)
    grown = 0
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g["ng"], ng[g["idx"]])
        assert np.array_equal(g["nr"], nr[g["idx"]])
        assert np.array_equal(g["labels"], pi)
        assert np.array_equal(g["pid"], pl["id"]) and np.array_equal(g["poff"], pl["offset"])
        assert np.array_equal(g["pidx"], pl["point_idx"]) and np.array_equal(g["pcenter"], pl["center"])
        assert np.array_equal(g["pnormal"].view(np.int64), pl["normal"].view(np.int64))
        assert int(g["components"]) == 6
        if world > 1:
            assert g["n_local"] < len(xyz) and 0 < int(g["n_grow"]) < len(xyz)  # both ranks grew some of the boxes
        seen[g["idx"]] += 1
        grown += int(g["n_grow"])
    assert (seen == 1).all() and grown == len(xyz)


def test_two_process_sharded_run_on_one_gpu(oracle, tmp_path):
    world = 2
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, "gloo"), nprocs=world, join=True)
    _check(oracle, out, world)


def test_rccl_collectives_at_world_one(oracle, tmp_path):
    """The nccl (= RCCL) branch of every collective the sharded path issues -- all-reduce MIN / MAX / SUM,
    all-gather, all-to-all with split sizes, all on DEVICE tensors -- executed at world size 1, the only size a
    one-GPU box can host (two RCCL ranks cannot share a device).  Results must equal the oracle's."""
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(1, _free_port(), out, "nccl"), nprocs=1, join=True)
    _check(oracle, out, 1)


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
    assert d["config"]["points_total"] == 99946 and "connected components" in d["config"]["parallelism"]
    assert {"partition_ms", "halo_ms", "knn_normals_ms", "components_ms", "redistribute_ms", "grow_ms", "labels_ms"} <= set(d["stages_ms"])
    assert d["roofline"]["kernel"] and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["cores"] == 1
    # a mislabelled run is refused: WORLD_SIZE (1 rank) != --gpus
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert q.returncode == 2 and "refusing" in q.stderr
