"""Device building blocks of the component-sharded stage 3 (csrc/bs_shard.hip) through the C ABI, each against an
independent CPU computation: bs_cc_hook_dev vs scipy's connected components, bs_owner_fetch_dev /
bs_plane_seeds_dev / bs_labels_from_owner_dev vs the CPU oracle's owners, bs_remap_rows_dev vs numpy, and
bs_region_grow_dev on foreign buffers (the grower builds its own Morton order) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def test_cc_hook_matches_scipy_components(gpu_ctx, oracle):
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    from buildingsegment_amd import synth
    torch = _torch()
    xyz = synth.boxes()
    n = len(xyz)
    ng, _ = oracle.knn_normals(xyz, k=15, want_normals=False)
    A = sp.coo_matrix((np.ones(ng.size), (np.repeat(np.arange(n), 15), ng.ravel())), shape=(n, n))
    ncomp, lab = cg.connected_components(A, directed=False)
    want = np.zeros(n, np.int64)
    for c in range(ncomp):  # root = smallest index of the component
        m = np.flatnonzero(lab == c)
        want[m] = m.min()
    dev = torch.device("cuda", 0)
    # split the rows in two "ranks" that hook into ONE parent array one after the other, then once more: no hooks left
    d_rows = torch.from_numpy(ng).to(dev)
    d_gidx = torch.arange(n, dtype=torch.int32, device=dev)
    parent = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    half = n // 2
    h1 = gpu_ctx.cc_hook_dev(d_rows[:half].data_ptr(), d_gidx[:half].data_ptr(), half, 15, parent.data_ptr(), n)
    h2 = gpu_ctx.cc_hook_dev(d_rows[half:].contiguous().data_ptr(), d_gidx[half:].contiguous().data_ptr(), n - half, 15,
                             parent.data_ptr(), n)
    h3 = gpu_ctx.cc_hook_dev(d_rows.data_ptr(), d_gidx.data_ptr(), n, 15, parent.data_ptr(), n)
    assert h1 > 0 and h2 > 0 and h3 == 0
    assert h1 + h2 == n - ncomp  # a forest over n nodes with ncomp trees
    assert np.array_equal(parent.cpu().numpy().astype(np.int64), want)
    # identity gidx may be passed as NULL
    parent2 = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.cc_hook_dev(d_rows.data_ptr(), 0, n, 15, parent2.data_ptr(), n)
    assert np.array_equal(parent2.cpu().numpy().astype(np.int64), want)


def test_owner_seeds_and_labels_match_the_oracle(gpu_ctx, oracle):
    from buildingsegment_amd import api, synth
    torch = _torch()
    xyz = synth.boxes()
    n = len(xyz)
    ng, nr = oracle.knn_normals(xyz, k=15)
    pi, pl, ow = oracle.region_grow(xyz, nr, ng, want_owner=True)
    seeds = pl["point_idx"][pl["offset"][:-1]].astype(np.int32)
    dev = torch.device("cuda", 0)
    d_xyz, d_nr, d_ng = (torch.from_numpy(a).to(dev) for a in (xyz, nr, ng))
    d_lab = torch.empty(n, dtype=torch.int32, device=dev)
    d_own = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    p = api.default_params(k=15)
    gpu_ctx.region_grow_dev(d_xyz.data_ptr(), d_nr.data_ptr(), d_ng.data_ptr(), n, d_lab.data_ptr(), p)
    gpu_ctx.owner_fetch_dev(d_own.data_ptr())
    npl = gpu_ctx.plane_seeds_dev(0, 0)
    d_seeds = torch.empty(npl, dtype=torch.int32, device=dev)
    assert gpu_ctx.plane_seeds_dev(d_seeds.data_ptr(), npl) == npl
    gpu_ctx.sync()
    assert np.array_equal(d_lab.cpu().numpy(), pi)
    assert np.array_equal(d_seeds.cpu().numpy(), seeds)
    # owners: equal wherever a point is labelled; the oracle's owner of an unlabelled point is -1 as well
    assert np.array_equal(d_own.cpu().numpy(), ow)
    d_lab2 = torch.empty(n, dtype=torch.int32, device=dev)
    gpu_ctx.labels_from_owner_dev(d_own.data_ptr(), n, d_seeds.data_ptr(), npl, d_lab2.data_ptr())
    gpu_ctx.sync()
    assert np.array_equal(d_lab2.cpu().numpy(), pi)  # planeIdx = 1 + #(committed seeds < owner)
    # the single-wave grower keeps no owners: refused, not garbage
    gpu_ctx.region_grow_dev(d_xyz.data_ptr(), d_nr.data_ptr(), d_ng.data_ptr(), n, d_lab.data_ptr(), api.default_params(k=15, rg_mode=1))
    with pytest.raises(api.BsError):
        gpu_ctx.owner_fetch_dev(d_own.data_ptr())


def test_remap_rows(gpu_ctx):
    torch = _torch()
    rng = np.random.default_rng(5)
    sg = np.sort(rng.choice(1_000_000, 50_000, replace=False)).astype(np.int32)
    rows = sg[rng.integers(0, len(sg), (20_000, 16))]
    dev = torch.device("cuda", 0)
    d_sg, d_rows = torch.from_numpy(sg).to(dev), torch.from_numpy(rows).to(dev)
    d_out = torch.empty_like(d_rows)
    torch.cuda.synchronize()
    assert gpu_ctx.remap_rows_dev(d_rows.data_ptr(), 20_000, 16, d_sg.data_ptr(), len(sg), d_out.data_ptr()) == 0
    assert np.array_equal(sg[d_out.cpu().numpy()], rows)
    rows[7, 3] = sg[100] + 1 if sg[100] + 1 != sg[101] else -5  # an index that is not in the local cloud
    d_rows = torch.from_numpy(rows).to(dev)
    torch.cuda.synchronize()
    assert gpu_ctx.remap_rows_dev(d_rows.data_ptr(), 20_000, 16, d_sg.data_ptr(), len(sg), d_out.data_ptr()) != 0


def test_foreign_buffers_grow_in_morton_order_and_match_the_fused_path(gpu_ctx, oracle):
    """bs_region_grow_dev on buffers it has never seen (what a rank of the sharded stage 3 calls): same labels and
    planes as the fused pipeline and as the oracle, with the Morton order built by the grower itself."""
    from buildingsegment_amd import api, synth
    torch = _torch()
    xyz = synth.urban(400_000, seed=12)
    n = len(xyz)
    p = api.default_params(k=16)
    neigh, normals, plane_idx, planes = gpu_ctx.segment(xyz, p)
    dev = torch.device("cuda", 0)
    d_xyz, d_nr, d_ng = (torch.from_numpy(a).to(dev) for a in (xyz, normals, neigh))
    d_lab = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.region_grow_dev(d_xyz.data_ptr(), d_nr.data_ptr(), d_ng.data_ptr(), n, d_lab.data_ptr(), p)
    t_foreign = gpu_ctx.timings()
    assert np.array_equal(d_lab.cpu().numpy(), plane_idx)
    pl2 = gpu_ctx.planes_fetch()
    assert len(pl2) == len(planes) and all(np.array_equal(a.pointIdx, b.pointIdx) for a, b in zip(planes, pl2))
    opi, _ = oracle.region_grow(xyz, normals, neigh)
    assert np.array_equal(plane_idx, opi)
    assert t_foreign["grow_ms"] > 0
