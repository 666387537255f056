"""CPU oracle, stages 1-2 (kNN + hybrid-radius PCA normals).  PARITY UNPINNED
at the Open3D boundary (Open3D 0.19.0 is not vendored in the reference and is
absent here; the reference has no golden vectors): the restatement is checked
against independent implementations instead -- brute force, scipy's cKDTree and
numpy's symmetric eigensolver."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from buildingsegment_amd import synth


def _d2(xyz, neigh):
    d = xyz[neigh].astype(np.int64) - xyz[:, None, :].astype(np.int64)
    return (d * d).sum(-1)


@pytest.mark.parametrize("k", [1, 2, 15, 16, 32])
def test_grid_knn_equals_brute_force(oracle, k):
    xyz = synth.uniform(3000, seed=3)
    a = oracle.knn_normals(xyz, k=k, want_normals=False)[0]
    b = oracle.knn_brute(xyz, k=k)
    assert np.array_equal(a, b)


def test_canonical_order_d2_then_index(oracle):
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 6, (400, 3)).astype(np.int32) * 10  # many exact ties + duplicates
    ng = oracle.knn_normals(xyz, k=15, want_normals=False)[0]
    d2 = _d2(xyz, ng)
    assert (np.diff(d2, axis=1) >= 0).all()
    tie = np.diff(d2, axis=1) == 0
    assert (np.diff(ng, axis=1)[tie] > 0).all()
    assert np.array_equal(ng, oracle.knn_brute(xyz, k=15))


def test_knn_distances_match_ckdtree(oracle):
    xyz = synth.plane_cube()[:20000].copy()
    ng = oracle.knn_normals(xyz, k=16, want_normals=False)[0]
    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=16)
    assert np.array_equal(_d2(xyz, ng), np.rint(d * d).astype(np.int64))
    assert (ng[:, 0] == np.arange(len(xyz))).all()  # no duplicate points here: slot 0 is self


def test_n_equals_k_and_n_less_than_k(oracle):
    xyz = synth.uniform(15, seed=2)
    ng = oracle.knn_normals(xyz, k=15, want_normals=False)[0]
    assert sorted(ng[3]) == list(range(15))
    with pytest.raises(ValueError):
        oracle.knn_normals(xyz, k=16)


def test_explicit_cell_sizes_agree(oracle):
    xyz = synth.facade(n_side=120, seed=4)
    a = oracle.knn_normals(xyz, k=16, cell=0)
    for cell in (37, 100, 450, 5000):
        b = oracle.knn_normals(xyz, k=16, cell=cell)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_hybrid_neighbourhood_and_normals_vs_eigh(oracle):
    xyz = synth.plane_cube()[:15000].copy()
    _, nr = oracle.knn_normals(xyz, k=15, radius=100.0, max_nn=50)
    tree = cKDTree(xyz.astype(np.float64))
    checked = 0
    for i in range(0, len(xyz), 37):
        d, idx = tree.query(xyz[i].astype(np.float64), k=50)
        idx = idx[d * d < 100.0 ** 2 - 0.5]
        if len(idx) < 3:
            assert tuple(nr[i]) == (0.0, 0.0, 1.0)
            continue
        w, v = np.linalg.eigh(np.cov(xyz[idx].astype(np.float64).T, bias=True))
        if w[1] - w[0] < 1e-3 * max(w[2], 1.0):
            continue
        assert 1.0 - abs(float(v[:, 0] @ nr[i])) < 1e-9
        assert nr[i, 2] >= 0.0 and abs(np.linalg.norm(nr[i]) - 1.0) < 1e-12
        checked += 1
    assert checked > 200


def test_max_nn_cut_takes_the_nearest(oracle):
    rng = np.random.default_rng(9)
    xyz = rng.integers(0, 120, (4000, 3)).astype(np.int32)  # ~2300 points per r=100 ball
    _, nr = oracle.knn_normals(xyz, k=15, radius=100.0, max_nn=50)
    ng50 = oracle.knn_brute(xyz, k=50)
    for i in (0, 17, 999):
        assert np.array_equal(oracle.normal_from_list(xyz, ng50[i]), nr[i])


def test_degenerate_neighbourhoods(oracle):
    xyz = np.array([[0, 0, 0], [5000, 0, 0], [0, 5000, 0], [5000, 5000, 0], [2500, 2500, 9000]], np.int32)
    _, nr = oracle.knn_normals(xyz, k=3)
    assert (nr == np.array([0.0, 0.0, 1.0])).all()  # < 3 neighbours inside r -> identity covariance
    line = np.stack([np.arange(50) * 10, np.zeros(50), np.zeros(50)], 1).astype(np.int32)
    _, nl = oracle.knn_normals(line, k=3)
    assert np.isfinite(nl).all() and (np.abs(np.linalg.norm(nl, axis=1) - 1) < 1e-12).all()
    assert (np.abs(nl[:, 0]) < 1e-12).all()  # normal is orthogonal to the line
    assert np.array_equal(oracle.fast_eigen3x3([0, 0, 0, 0, 0, 0]), [0.0, 0.0, 0.0])
    assert np.array_equal(oracle.fast_eigen3x3([2, 0, 0, 1, 0, 3]), [0.0, 1.0, 0.0])
