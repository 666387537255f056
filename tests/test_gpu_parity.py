"""GPU parity tests proper: the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs.  Bit-exact for neighbour indices, labels,
plane lists, centres; normals bit-exact as well (shared deterministic acos/cos,
contraction off) -- the north-star tolerance of 1e-5 is asserted first and the
exact comparison second so a failure says which bar was missed."""
import numpy as np
import pytest

from buildingsegment_amd import api, synth

pytestmark = pytest.mark.gpu


def _check_knn_normals(ctx, O, xyz, k, **kw):
    p = api.default_params(k=k, **kw)
    neigh, normals = ctx.knn_normals(xyz, p)
    oneigh, onormals = O.knn_normals(xyz, k=k, radius=p.radius, max_nn=p.max_nn)
    assert np.array_equal(neigh, oneigh), f"{(neigh != oneigh).any(axis=1).sum()} rows differ"
    assert np.abs(normals - onormals).max() <= 1e-5  # north-star tolerance
    assert np.array_equal(normals, onormals)
    return neigh, normals


def _check_grow(ctx, O, xyz, normals, neigh, **kw):
    p = api.default_params(k=neigh.shape[1], **kw)
    plane_idx, planes = ctx.region_grow(xyz, normals, neigh, p)
    opi, opl = O.region_grow(xyz, normals, neigh, th_thickness=p.th_thickness,
                             th_point_count=p.th_point_count, cos_th=p.cos_th)
    assert np.array_equal(plane_idx, opi), f"{(plane_idx != opi).sum()} labels differ"
    assert len(planes) == len(opl["id"])
    for i, pl in enumerate(planes):
        assert pl.id == opl["id"][i]
        assert np.array_equal(pl.pointIdx, opl["point_idx"][opl["offset"][i]:opl["offset"][i + 1]])
        assert np.array_equal(pl.center, opl["center"][i])
        assert np.array_equal(pl.normal, opl["normal"][i])
    return plane_idx, planes


def test_plane_cube_100k_reference_literals(gpu_ctx, oracle):
    xyz = synth.plane_cube()
    neigh, normals = _check_knn_normals(gpu_ctx, oracle, xyz, 15)
    plane_idx, planes = _check_grow(gpu_ctx, oracle, xyz, normals, neigh)
    assert len(planes) >= 1 and (plane_idx > 0).sum() > 60000


@pytest.mark.parametrize("k", [2, 8, 16, 17, 32])
def test_knn_k_sweep_uniform(gpu_ctx, oracle, k):
    xyz = synth.uniform(30000, seed=11)
    _check_knn_normals(gpu_ctx, oracle, xyz, k)


def test_facade_200k_k16(gpu_ctx, oracle):
    xyz = synth.facade(n_side=450, seed=2)
    neigh, normals = _check_knn_normals(gpu_ctx, oracle, xyz, 16)
    _check_grow(gpu_ctx, oracle, xyz, normals, neigh)


def test_segment_fused_matches_stages(gpu_ctx, oracle):
    xyz = synth.urban(150_000, seed=5)
    p = api.default_params(k=16)
    neigh, normals, plane_idx, planes = gpu_ctx.segment(xyz, p)
    oneigh, onormals = oracle.knn_normals(xyz, k=16)
    assert np.array_equal(neigh, oneigh) and np.array_equal(normals, onormals)
    opi, opl = oracle.region_grow(xyz, onormals, oneigh)
    assert np.array_equal(plane_idx, opi)
    assert sum(len(p.pointIdx) for p in planes) == len(opl["point_idx"])
