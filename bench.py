#!/usr/bin/env python3
"""bench.py -- Mpoints/s segmented (kNN + normal + label) on MI355X.

One "step" = one pass of the whole hot path (search grid, kNN + PCA normals,
region-growing labels) over one device-resident synthetic cloud.

N = 1 (default): the configuration north_star quotes its target on -- the
50 M-point synthetic building cloud at k=16 on ONE GPU (`urban_50m`, SURVEY.md
8(d) C3; it fits one MI355X).  `secondary` carries BASELINE.json configs[1]
(1 M facade, k=16) and configs[2] (10 M urban block, k=32), measured live with
the same protocol, each with its own `roofline` block.

N > 1 (`--gpus N`, one rank per GPU over RCCL): ONE urban_50m cloud is segmented
by buildingsegment_amd.dist.segment_sharded_dev -- stages 1-2 sharded by Morton
slabs with a device-resident halo exchange, stage 3 sharded exactly by connected
components of the kNN graph (union-find all-reduce, per-rank growth, global plane
ids from the all-gathered committed seeds) -- strong scaling, stage times reported
separately.
`python bench.py --gpus N` without a launcher starts the N ranks itself (before
anything touches the GPU); under torchrun WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0): the driver's contract fields plus `roofline`
(dominant kernel: algorithmic bytes / HIP-event launch time on the launch stream)
and `cpu_baseline` (the CPU oracle timed on this box's host cores by rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOADS = ["urban_50m", "urban_10m", "facade_1m", "plane_cube_100k", "uniform_1m", "uniform_10m", "urban_2m", "urban_200m"]
K_DEFAULT = {"facade_1m": 16, "urban_10m": 32, "urban_50m": 16, "urban_2m": 16, "plane_cube_100k": 15,
             "uniform_1m": 16, "uniform_10m": 16, "urban_200m": 16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="urban_50m", choices=WORKLOADS)
    ap.add_argument("--k", type=int, default=0, help="neighbour-list length (0 = workload default)")
    ap.add_argument("--rg-mode", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-audit", action="store_true",
                    help="skip the untimed extra pass that replays every plane attempt against the final owners")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 path "
                         "with several ranks sharing one GPU)")
    ap.add_argument("--secondary", default="facade_1m,urban_10m,uniform_1m,uniform_10m",
                    help="comma-separated workloads measured live after the headline and reported under "
                         "'secondary' ('' = none; ignored for N>1)")
    ap.add_argument("--concurrent", type=int, default=2,
                    help="N=1 only: also report the aggregate throughput of this many independent copies of the workload "
                         "segmented at the same time on the one GPU ('concurrent_clouds'; 0/1 = skip)")
    ap.add_argument("--replicas", action="store_true",
                    help="N>1: every rank segments its own copy of the workload (weak scaling of independent "
                         "clouds, no data-path collective) instead of sharding one cloud")
    return ap.parse_args()


def make_cloud(name: str, rank: int = 0):
    """The named BASELINE workload: (xyz int32 [n,3], default k)."""
    from buildingsegment_amd import synth
    cache = os.environ.get("BS_CLOUD_CACHE")  # developer aid: reuse a generated cloud between processes on one box
    if cache:
        f = os.path.join(cache, f"bs_cloud_{name}.npy")
        if os.path.exists(f):
            return np.load(f), K_DEFAULT[name]
        os.environ.pop("BS_CLOUD_CACHE")
        xyz, k = make_cloud(name, rank)
        os.environ["BS_CLOUD_CACHE"] = cache
        np.save(f + ".tmp.npy", xyz)
        os.replace(f + ".tmp.npy", f)
        return xyz, k
    if name == "facade_1m":
        return synth.facade(n_side=1000, seed=2), 16
    if name == "urban_10m":
        return synth.urban(10_000_000, seed=3), 32
    if name == "urban_50m":
        return synth.urban(50_000_000, seed=4), 16
    if name == "urban_200m":
        return synth.urban(200_000_000, seed=5), 16
    if name == "urban_2m":
        return synth.urban(2_000_000, seed=3), 16
    if name == "plane_cube_100k":
        return synth.plane_cube(seed=1), 15
    if name == "uniform_1m":
        return synth.uniform(1_000_000, seed=6), 16
    if name == "uniform_10m":
        return synth.uniform(10_000_000, seed=6), 16
    raise ValueError(name)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(name: str, xyz: np.ndarray, k: int):
    """CPU oracle (oracle/bs_oracle.c, ONE thread) on a bounded sample of the same workload, with the per-stage split.

    Small workloads run whole.  The urban clouds are generated building by building from
    counter-based streams, so `synth.urban(m, seed)` IS the first m points (the first
    buildings) of the larger cloud of the same seed: that sub-scene is segmented end to end
    (kNN + normals + region grow) -- the same density, plane sizes and k as the full job.
    The uniform control cloud has one density at every size (mean spacing 50 mm): its 1 M instance stands for 10 M."""
    from buildingsegment_amd import synth
    from oracle import oracle as O
    n = len(xyz)
    sample_n = {"urban_50m": 2_000_000, "urban_200m": 2_000_000, "urban_10m": 1_000_000, "uniform_10m": 1_000_000}.get(name, n)
    if sample_n < n and name.startswith("urban"):
        seed = {"urban_50m": 4, "urban_200m": 5, "urban_10m": 3}[name]
        sub = synth.urban(sample_n, seed=seed)
        what = (f"synth.urban({sample_n}, seed={seed}) = the first {sample_n} points (first buildings) of the {name} "
                "generator stream, whole path")
    elif sample_n < n:
        sub = synth.uniform(sample_n, seed=6)
        what = f"synth.uniform({sample_n}, seed=6): the same density (mean spacing 50 mm) as {name}, whole path"
    else:
        sub = xyz
        what = f"whole workload ({n} points)"
    t0 = time.perf_counter()
    neigh, normals = O.knn_normals(sub, k=k)
    t1 = time.perf_counter()
    O.region_grow(sub, normals, neigh)
    t2 = time.perf_counter()
    m = len(sub)
    return {"value": m / (t2 - t0) / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "stages_s": {"knn_normals": t1 - t0, "region_grow": t2 - t1},
            "stage_Mpoints_per_s": {"knn_normals": m / (t1 - t0) / 1e6, "region_grow": m / max(t2 - t1, 1e-9) / 1e6},
            "sample": f"{what}: kNN+normals {t1 - t0:.2f}s, region grow {t2 - t1:.2f}s on 1 thread"}


def roofline_block(n, k, stage, launches_per_step, rg_mode, workload):
    """Dominant kernel by time.  Algorithmic bytes per point:
      kNN+normals: read xyz 12 + write k*4 + write normal 24 (+12: second read of xyz)
      region grow: read neigh row k*4 + xyz 12 + normal 24 + write label 4
    One pass of region growing is spread over `launches_per_step` launches of the
    plane-growth kernel (one per speculative round): bytes per launch = n(4k+40)/rounds,
    average launch duration from HIP events recorded around each launch on the stream."""
    lps = max(launches_per_step, 1.0)
    grow_bytes = n * (4 * k + 40) / lps
    knn_bytes = n * (4 * k + 36 + 12)
    grow_avg = stage["grow_kernel_ms"] / lps
    if stage["grow_kernel_ms"] >= stage["knn_ms"]:
        # (the plane-growth launch has two step engines, picked per round: rocprof lists them as two kernels whose
        # launches together are the `launches_per_step` of this block)
        dom = "grow_spec_kernel + grow_spec2_kernel" if rg_mode != 1 else "grow_seq_kernel"
        dbytes, dms = grow_bytes, grow_avg
    else:
        dom, dbytes, dms = "knn_fast_kernel", knn_bytes, stage["knn_ms"]
    achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    # HBM bytes per launch of the dominant kernel: NOT live (the counters cannot be read from inside the
    # bench) -- taken from the committed rocprofv3 PMC passes of the same command (tools/pmc_traffic.py)
    traffic, src = None, None
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pt.get(workload) if isinstance(pt.get(workload), dict) else (pt if pt.get("workload") == workload else None)
        if ent and rg_mode != 1:
            if dom.startswith("grow_spec"):
                traffic = ent["grow_spec_kernel_bytes_per_call"] / lps
            else:
                traffic = ent["knn_fast_kernel_bytes_per_call"]
            src = "profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)"
    except (OSError, ValueError, KeyError, TypeError):
        traffic = None
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "alg_bytes_per_launch": dbytes, "avg_ms": dms, "launches_per_step": lps}


def measure_single(ctx, api, torch, dev, name, k_override, rg_mode, steps, warmup, fence, world=1, all_reduce_max=None,
                   audit=False):
    """Whole path on one GPU per rank (bs_segment_dev), inputs resident in HBM."""
    xyz, k = make_cloud(name)
    if k_override:
        k = k_override
    n = len(xyz)
    params = api.default_params(k=k, rg_mode=rg_mode)
    d_xyz = torch.from_numpy(xyz).to(dev)
    d_neigh = torch.empty((n, k), dtype=torch.int32, device=dev)
    d_normals = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_plane = torch.empty((n,), dtype=torch.int32, device=dev)

    def step():
        ctx.segment_dev(d_xyz.data_ptr(), n, d_plane.data_ptr(), params, d_neigh.data_ptr(), d_normals.data_ptr())

    for _ in range(warmup):
        step()
    fence()
    stage = {"grid_ms": 0.0, "knn_ms": 0.0, "grow_ms": 0.0, "grow_kernel_ms": 0.0, "grow_setup_ms": 0.0}
    launches = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        tm = ctx.timings()  # HIP-event stage times of this step (events on the launch stream)
        for kk in stage:
            stage[kk] += tm.get(kk, 0.0)
        launches += tm["grow_kernel_launches"]
    fence()
    elapsed = time.perf_counter() - t0
    if all_reduce_max is not None:
        elapsed = all_reduce_max(elapsed)
    tm = ctx.timings()
    for kk in stage:
        stage[kk] /= max(steps, 1)
    audit_res = None
    if audit:  # one more, UNTIMED pass with the replay certificate switched on (bs_set_audit)
        ctx.set_audit(True)
        try:
            step()
            ta = ctx.timings()
        finally:
            ctx.set_audit(False)
        audit_res = {"attempts_replayed": ta["audit_attempts"], "mismatches": ta["audit_mismatches"], "ms": ta["audit_ms"],
                     "note": "every plane attempt grown again against the final owners, lists / normals / centres compared "
                             "bit for bit; untimed extra pass"}
    res = {"workload": name, "points": n, "k": k, "steps": steps, "warmup": warmup,
           "value": world * n * steps / elapsed / 1e6, "unit": "Mpoints/s", "ms_per_step": elapsed / max(steps, 1) * 1e3,
           "stages_ms": stage, "rg_rounds": tm["rg_rounds"], "largest_plane": tm["largest_plane"],
           "seed_attempts": tm["n_seed_attempts"], "fallback_queries": tm["n_fallback_queries"],
           "validation_rejects": tm["validation_rejects"],
           "validation_rejects_by_check": {"robbed_after_finish": tm["rej_v1_robbed"], "entry_without_claim": tm["rej_v1_tag"],
                                           "duplicate_entry": tm["rej_v1_dup"], "state_not_reproducible": tm["rej_v3_state"]},
           "inconsistent_by_test": {"seed_row": tm["incons_seed"], "list": tm["incons_list"], "assumption_log": tm["incons_log"]},
           "tie_rows": tm["tie_rows"], "tie_rows_frac": tm["tie_rows"] / max(n, 1),
           "end_to_end_alg_GBps": n * (88 + 8 * k) / (elapsed / max(steps, 1)) / 1e9,
           "roofline": roofline_block(n, k, stage, launches / max(steps, 1), rg_mode, name),
           "radius_mm": params.radius, "max_nn": params.max_nn}
    if audit_res:
        res["audit"] = audit_res
    del d_xyz, d_neigh, d_normals, d_plane
    torch.cuda.empty_cache()
    return res, xyz, k


def measure_concurrent(api, torch, dev, name, k_override, rg_mode, steps, nctx):
    """NOT the headline: `nctx` independent copies of the workload segmented at the same time on ONE GPU, one
    context + stream + host thread each (ctypes releases the GIL; the speculative scheduler is host driven).
    A single pass is bounded by a chain of single-wave steps and leaves most of the chip idle; independent
    clouds (tiles of a scene) fill it.  Aggregate throughput = nctx * n * steps / wall."""
    import threading
    xyz, k = make_cloud(name)
    if k_override:
        k = k_override
    n = len(xyz)
    params = api.default_params(k=k, rg_mode=rg_mode)
    work = []
    for _ in range(nctx):
        c = api.Context(dev.index or 0)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            d_xyz = torch.from_numpy(xyz).to(dev)
            d_plane = torch.empty((n,), dtype=torch.int32, device=dev)
        work.append((c, st, d_xyz, d_plane))
    torch.cuda.synchronize(dev)

    def run(w, reps):
        c, _, d_xyz, d_plane = w
        for _ in range(reps):
            c.segment_dev(d_xyz.data_ptr(), n, d_plane.data_ptr(), params)

    for reps in (1, steps):  # warm-up, then the timed run
        ts = [threading.Thread(target=run, args=(w, reps)) for w in work]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    same = all(bool(torch.equal(work[0][3], w[3])) for w in work[1:])
    for c, *_ in work:
        c.close()
    del work
    torch.cuda.empty_cache()
    return {"contexts": nctx, "workload": name, "value": nctx * n * steps / el / 1e6, "unit": "Mpoints/s",
            "ms_per_pass_per_context": el / max(steps, 1) * 1e3, "labels_identical": same,
            "note": "independent copies of the workload on one GPU at the same time (one context, stream and host "
                    "thread each); aggregate throughput, NOT the headline metric"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a
    CHILD process (nothing has touched the GPU yet) and exit with its code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args))
        world = 1
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":  # rehearsal of the N>1 path on a box with fewer GPUs than ranks
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

    ctx = api.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def all_reduce_max(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    common = {"metric": "Mpoints/s segmented (kNN+normal+label)", "unit": "Mpoints/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
              "dtype": "int32 coordinates / f64 normals", "data": "synthetic"}
    out = None
    if world == 1 or args.replicas:
        res, xyz, k = measure_single(ctx, api, torch, dev, args.workload, args.k, args.rg_mode, args.steps, args.warmup,
                                     fence, world, all_reduce_max if world > 1 else None,
                                     audit=(world == 1 and not args.no_audit and args.rg_mode in (0, 2)))
        if rank == 0:
            out = dict(common)
            out.update({"value": res["value"], "ms_per_step": res["ms_per_step"], "scaling": "weak",
                        "config": {"workload": args.workload, "points_per_gpu": res["points"], "k": res["k"],
                                   "radius_mm": res["radius_mm"], "max_nn": res["max_nn"], "rg_mode": args.rg_mode,
                                   "largest_plane": res["largest_plane"], "seed_attempts": res["seed_attempts"],
                                   "fallback_queries": res["fallback_queries"], "rg_rounds": res["rg_rounds"],
                                   "validation_rejects": res["validation_rejects"],
                                   "validation_rejects_by_check": res["validation_rejects_by_check"],
                                   "tie_rows": res["tie_rows"], "tie_rows_frac": res["tie_rows_frac"],
                                   "parallelism": "1 GPU, whole path" if world == 1 else f"{world} independent replicas"},
                        "stages_ms": res["stages_ms"], "end_to_end_alg_GBps": res["end_to_end_alg_GBps"],
                        "roofline": res["roofline"]})
            if "audit" in res:
                out["audit"] = res["audit"]
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.workload, xyz, k)
        del xyz
        if world == 1 and args.secondary:
            sec = []
            for name in [s for s in args.secondary.split(",") if s and s != args.workload]:
                r2, xyz2, k2 = measure_single(ctx, api, torch, dev, name, 0, args.rg_mode, 2, 1, fence,
                                              audit=(not args.no_audit and args.rg_mode in (0, 2)))
                for drop in ("radius_mm", "max_nn"):
                    r2.pop(drop)
                if not args.no_cpu_baseline:  # each secondary beside ITS OWN one-thread CPU oracle run (per-stage split)
                    r2["cpu_baseline"] = cpu_baseline(name, xyz2, k2)
                del xyz2
                sec.append(r2)
            if rank == 0 and sec:
                out["secondary"] = sec
        if world == 1 and args.concurrent > 1:
            out["concurrent_clouds"] = measure_concurrent(api, torch, dev, args.workload, args.k, args.rg_mode, args.steps,
                                                          args.concurrent)
    else:
        # ONE cloud sharded over the ranks (north_star: Morton slabs + halo exchange + label union-find all-reduce):
        # stages 1-2 by Morton slabs, stage 3 by connected components of the kNN graph (exact; dist.py)
        from buildingsegment_amd import dist as bsd
        xyz, k = make_cloud(args.workload)
        if args.k:
            k = args.k
        n = len(xyz)
        params = api.default_params(k=k, rg_mode=args.rg_mode)
        b = bsd.slab_bounds(n, world)
        # resident input: rank r holds the r-th 1/N of the cloud in INPUT order (+ the global indices);
        # the Morton partition, the halo exchange and every other exchange are inside the timed region
        d_own = torch.from_numpy(xyz[b[rank]:b[rank + 1]]).to(dev)
        d_gidx = torch.arange(b[rank], b[rank + 1], dtype=torch.int32, device=dev)
        if rank != 0 or args.no_cpu_baseline:
            del xyz
        keys = ["partition_ms", "halo_ms", "knn_normals_ms", "components_ms", "redistribute_ms", "localize_ms", "grow_ms",
                "labels_ms"]
        st = {kk: 0.0 for kk in keys}
        kst = {"grid_ms": 0.0, "knn_ms": 0.0, "grow_kernel_ms": 0.0, "grow_setup_ms": 0.0}
        launches = 0
        info = {}
        for _ in range(args.warmup):
            bsd.segment_sharded_dev(ctx, d_own, d_gidx, n, params)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, info = bsd.segment_sharded_dev(ctx, d_own, d_gidx, n, params)
            for kk in st:
                st[kk] += info["stage_ms"][kk]
            tm = ctx.timings()  # HIP-event times of this rank's kNN and growth kernels in this step
            for kk in kst:
                kst[kk] += tm.get(kk, 0.0)
            launches += tm["grow_kernel_launches"]
        fence()
        elapsed = all_reduce_max(time.perf_counter() - t0)
        for kk in st:
            st[kk] = all_reduce_max(st[kk] / max(args.steps, 1))
        n_grow_max = int(all_reduce_max(float(info.get("n_grow", 0))))
        if rank == 0:
            for kk in kst:
                kst[kk] /= max(args.steps, 1)
            # roofline of rank 0's dominant kernel: its own share of the work (n_own queries / n_grow grown points)
            stage0 = {"knn_ms": kst["knn_ms"], "grow_kernel_ms": kst["grow_kernel_ms"]}
            n_knn, n_grow = int(info.get("n_own", 0)), int(info.get("n_grow", 0))
            lps = launches / max(args.steps, 1)
            if kst["grow_kernel_ms"] >= kst["knn_ms"] and n_grow:
                roof = roofline_block(n_grow, k, stage0, lps, args.rg_mode, "")
            else:
                roof = roofline_block(n_knn, k, {"knn_ms": kst["knn_ms"], "grow_kernel_ms": 0.0}, 1.0, args.rg_mode, "")
            roof["scope"] = f"rank 0 of {world}: {n_knn} kNN queries, {n_grow} points grown"
            out = dict(common)
            out.update({"value": n * args.steps / elapsed / 1e6, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                        "scaling": "strong",
                        "config": {"workload": args.workload, "points_total": n, "k": k, "rg_mode": args.rg_mode,
                                   "parallelism": f"stages 1-2: {world} Morton slabs, partition + halo by device all-to-all ({args.backend}); "
                                                  "stage 3: connected components of the kNN graph dealt to the ranks (union-find "
                                                  "all-reduce MIN, one all-to-all per array), each rank grows whole components, "
                                                  "plane ids from the all-gathered committed seeds, label all-reduce MAX",
                                   "halo_mm": info.get("halo"), "halo_retries": info.get("retries"),
                                   "n_local_rank0": info.get("n_local"), "components": info.get("components"),
                                   "cc_iterations": info.get("cc_iterations"), "points_grown_rank0": n_grow,
                                   "points_grown_max_rank": n_grow_max, "planes": info.get("n_planes_total")},
                        "stages_ms": st, "rank0_kernel_ms": kst,
                        "stage12_Mpoints_per_s": n / ((st["partition_ms"] + st["halo_ms"] + st["knn_normals_ms"]) * 1e-3) / 1e6,
                        "roofline": roof})
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.workload, xyz, k)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
