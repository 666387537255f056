#!/usr/bin/env python3
"""bench.py -- Mpoints/s segmented (kNN + normal + label) on MI355X.

One "step" = one pass of the whole hot path (bs_segment_dev: search grid,
kNN + PCA normals, region-growing labels) over one device-resident synthetic
cloud.  N=1 workload: BASELINE.json configs[1], the 1 M-point synthetic
building facade at k=16 (buildingsegment_amd.synth.facade).
With --gpus N every rank segments its own facade (independent objects, no
data-path collective): weak scaling.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
``roofline`` (dominant kernel, HIP-event timed on the launch stream),
``cpu_baseline`` (the CPU oracle timed on this box's host cores, N=1 only) and
``secondary`` (configs[2], the 10 M-point urban block at k=32, measured live the
same way: the multi-plane regime; `--secondary ''` skips it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="facade_1m", choices=["facade_1m", "urban_10m", "plane_cube_100k",
                                                                "uniform_1m", "urban_2m", "urban_50m"])
    ap.add_argument("--k", type=int, default=0, help="neighbour-list length (0 = workload default)")
    ap.add_argument("--rg-mode", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 control "
                         "flow with several ranks sharing one GPU)")
    ap.add_argument("--secondary", default="urban_10m",
                    help="second workload measured live and reported under 'secondary' ('' = none)")
    return ap.parse_args()


def make_cloud(name: str, rank: int):
    """The named BASELINE workload.  Every rank segments its own copy of the SAME cloud (the
    config names one seed): weak scaling then measures the system, not the luck of a seed --
    the façade's critical chain varies by 40 % between seeds."""
    from buildingsegment_amd import synth
    if name == "facade_1m":
        return synth.facade(n_side=1000, seed=2), 16
    if name == "urban_10m":
        return synth.urban(10_000_000, seed=3), 32
    if name == "urban_50m":
        return synth.urban(50_000_000, seed=4), 16
    if name == "urban_2m":
        return synth.urban(2_000_000, seed=3), 16
    if name == "plane_cube_100k":
        return synth.plane_cube(seed=1), 15
    if name == "uniform_1m":
        return synth.uniform(1_000_000, seed=6), 16
    raise ValueError(name)


def cpu_baseline(xyz: np.ndarray, k: int):
    """CPU oracle (oracle/, single thread) on a bounded sample of the same
    workload: kNN + normals for the first q points against the full cloud,
    region growing on the first min(n, 1M)-point prefix cloud of its own."""
    from oracle import oracle as O
    n = len(xyz)
    # ~0.3 Mpts/s for stage 1-2 on one core: cap the sample at 2 M queries
    q = min(n, 2_000_000)
    t0 = time.perf_counter()
    neigh, normals = O.knn_normals(xyz, k=k, q0=0, q1=q)
    t1 = time.perf_counter()
    if q == n:
        O.region_grow(xyz, normals, neigh)
        t2 = time.perf_counter()
        t_total = t2 - t0
        sample = f"whole workload ({n} points): kNN+normals {t1 - t0:.2f}s, region grow {t2 - t1:.2f}s"
    else:
        # region grow needs the full graph; time it on an independent prefix cloud
        sub = np.ascontiguousarray(xyz[:q])
        ng2, nr2 = O.knn_normals(sub, k=k)
        t2 = time.perf_counter()
        O.region_grow(sub, nr2, ng2)
        t3 = time.perf_counter()
        t_total = (t1 - t0) + (t3 - t2)
        sample = (f"{q} of {n} queries against the full cloud for kNN+normals ({t1 - t0:.2f}s) + region grow "
                  f"of a {q}-point prefix cloud ({t3 - t2:.2f}s)")
    return {"value": q / t_total / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port", "sample": sample}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":  # rehearsal of the N>1 control flow on a box with fewer GPUs than ranks
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from buildingsegment_amd import api

    xyz, k = make_cloud(args.workload, rank)
    if args.k:
        k = args.k
    n = len(xyz)
    params = api.default_params(k=k, rg_mode=args.rg_mode)
    ctx = api.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    d_xyz = torch.from_numpy(xyz).to(dev)
    d_neigh = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_normals = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_plane = torch.empty((n,), dtype=torch.int32, device=dev)

    def step():
        ctx.segment_dev(d_xyz.data_ptr(), n, d_plane.data_ptr(), params, d_neigh.data_ptr(), d_normals.data_ptr())

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    stage = {"grid_ms": 0.0, "knn_ms": 0.0, "grow_ms": 0.0, "grow_kernel_ms": 0.0}
    launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = ctx.timings()  # HIP-event stage times of this step (events on the launch stream)
        for kk in stage:
            stage[kk] += tm[kk]
        launches += tm["grow_kernel_launches"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = ctx.timings()
    steps = max(args.steps, 1)
    for kk in stage:
        stage[kk] /= steps

    if rank == 0:
        value = world * n * args.steps / elapsed / 1e6
        # Dominant kernel by time.  Algorithmic bytes per point:
        #   kNN+normals: read xyz 12 + write k*4 + write normal 24 (+12: second read of xyz)
        #   region grow: read neigh row k*4 + xyz 12 + normal 24 + write label 4
        # One pass of region growing is spread over `launches/steps` launches of the
        # plane-growth kernel (one per speculative round): bytes per launch = n(4k+40)/rounds,
        # average launch duration from HIP events recorded around each launch on the stream.
        lps = max(launches / steps, 1.0)
        grow_bytes = n * (4 * k + 40) / lps
        knn_bytes = n * (4 * k + 36 + 12)
        grow_avg = stage["grow_kernel_ms"] / lps
        if stage["grow_kernel_ms"] >= stage["knn_ms"]:
            dom = "grow_spec_kernel" if args.rg_mode != 1 else "grow_seq_kernel"
            dbytes, dms = grow_bytes, grow_avg
        else:
            dom, dbytes, dms = "knn_fast_kernel", knn_bytes, stage["knn_ms"]
        achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
        # (tools/pmc_traffic.py; the counters cannot be read live from inside the bench)
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == args.workload and args.rg_mode != 1:
                if dom == "grow_spec_kernel":
                    traffic = pt["grow_spec_kernel_bytes_per_call"] / lps
                else:
                    traffic = pt["knn_fast_kernel_bytes_per_call"]
        except (OSError, ValueError, KeyError, TypeError):
            traffic = None
        out = {
            "metric": "Mpoints/s segmented (kNN+normal+label)",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 coordinates / f64 normals",
            "data": "synthetic",
            "config": {"workload": args.workload, "points_per_gpu": n, "k": k, "radius_mm": params.radius,
                       "max_nn": params.max_nn, "rg_mode": args.rg_mode,
                       "largest_plane": tm["largest_plane"], "seed_attempts": tm["n_seed_attempts"],
                       "fallback_queries": tm["n_fallback_queries"], "rg_rounds": tm["rg_rounds"]},
            "stages_ms": stage,
            "end_to_end_alg_GBps": n * (88 + 8 * k) / (elapsed / steps) / 1e9,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": dbytes, "avg_ms": dms, "launches_per_step": lps},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(xyz, k)
    # Secondary workload, measured live with the same barrier/timing protocol: BASELINE.json
    # configs[2] (10 M-point urban block, k=32) shows the multi-plane regime of stage 3, which
    # the single-surface facade cannot (its large planes form one dependency chain).
    if args.secondary and args.secondary != args.workload:
        del d_xyz, d_neigh, d_normals, d_plane
        torch.cuda.empty_cache()
        xyz2, k2 = make_cloud(args.secondary, rank)
        n2 = len(xyz2)
        p2 = api.default_params(k=k2, rg_mode=args.rg_mode)
        e_xyz = torch.from_numpy(xyz2).to(dev)
        e_neigh = torch.empty((n2, k2), dtype=torch.int32, device=dev)
        e_normals = torch.empty((n2, 3), dtype=torch.float64, device=dev)
        e_plane = torch.empty((n2,), dtype=torch.int32, device=dev)

        def step2():
            ctx.segment_dev(e_xyz.data_ptr(), n2, e_plane.data_ptr(), p2, e_neigh.data_ptr(), e_normals.data_ptr())

        step2()
        fence()
        t0 = time.perf_counter()
        for _ in range(2):
            step2()
        fence()
        el2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        tm2 = ctx.timings()
        if rank == 0:
            out["secondary"] = {"workload": args.secondary, "points_per_gpu": n2, "k": k2, "steps": 2, "warmup": 1,
                                "value": world * n2 * 2 / el2 / 1e6, "unit": "Mpoints/s", "ms_per_step": el2 / 2 * 1e3,
                                "grid_ms": tm2["grid_ms"], "knn_ms": tm2["knn_ms"], "grow_ms": tm2["grow_ms"],
                                "rg_rounds": tm2["rg_rounds"], "largest_plane": tm2["largest_plane"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
