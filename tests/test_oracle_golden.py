"""Pins the CPU oracle's stage 3 (region growing) to the reference's own code.

tests/golden/*.npz hold inputs and the outputs of oracle/_ref/ref_stage3 -- the
reference's seg_plane::get_planes/Broad/set_plane_color compiled verbatim from
/root/reference/tmc3/my_function.{h,cpp} in the build container
(tests/golden/make_golden.py).  The oracle's restatement must reproduce every
label, every plane list (order + duplicates), centres and normals bit for bit.
"""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if not os.path.basename(p).startswith(("raster_", "ply_")))  # stage-3 fixtures (raster / PLY ones have their own tests)


def test_fixtures_present():
    assert len(GOLD) >= 6


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_region_grow_matches_reference(oracle, path):
    g = np.load(path)
    pi, pl = oracle.region_grow(g["xyz"], g["normals"], g["neigh"])
    assert np.array_equal(pi, g["plane_idx"])
    assert np.array_equal(pl["id"], g["id"])
    assert np.array_equal(pl["offset"], g["offset"])
    assert np.array_equal(pl["point_idx"], g["point_idx"])
    assert np.array_equal(pl["center"], g["center"])
    assert np.array_equal(pl["normal"], g["normal"])  # bit-exact f64


def test_p1_seed_duplicate_and_rollback(oracle):
    """SURVEY Appendix B.5 probe P1: one plane of 1601 entries (seed twice),
    centre (974,974,0), the perpendicular patch rolled back to -1."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_patch_p1.npz"))
    pi, pl = oracle.region_grow(g["xyz"], g["normals"], g["neigh"])
    assert len(pl["id"]) == 1 and pl["id"][0] == 1
    lst = pl["point_idx"]
    assert len(lst) == 1601 and (lst == lst[0]).sum() == 2
    assert tuple(pl["center"][0]) == (974, 974, 0)
    assert (pi[:1600] == 1).all() and (pi[1600:] == -1).all()


def test_p5_orphans(oracle):
    """Probe P5: a failed seed leaves its accepted neighbours labelled (Q2)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "orphans_p5.npz"))
    pi, pl = oracle.region_grow(g["xyz"], g["normals"], g["neigh"])
    inlist = np.zeros(len(pi), bool)
    inlist[pl["point_idx"]] = True
    orphans = (pi > 0) & ~inlist
    assert orphans.sum() == 13
    assert np.array_equal(pi, g["plane_idx"])


def test_colors_follow_glibc_rand(oracle):
    """set_plane_color (my_function.cpp:260-275) with glibc's unseeded rand():
    first plane colour (238,141,232) in the G,B,R slots."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_patch_p1.npz"))
    col = g["colors"]
    assert tuple(col[0]) == (238, 141, 232)
    assert (col[1600:] == 0).all()


def test_threshold_parameters_are_honoured(oracle):
    """Non-default thresholds (the reference hard-codes them) change results
    monotonically: a larger th_point_count can only drop planes."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "plane_cube_12k.npz"))
    _, pl_a = oracle.region_grow(g["xyz"], g["normals"], g["neigh"], th_point_count=400)
    _, pl_b = oracle.region_grow(g["xyz"], g["normals"], g["neigh"], th_point_count=750)
    assert len(pl_b["id"]) <= len(pl_a["id"])
    assert all(s > 750 for s in np.diff(pl_b["offset"]))
