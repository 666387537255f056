"""End-to-end drop-in check on the GPU: the tmc3-compatible CLI (host/tmc3:
PLY in -> bbox shift -> kNN+normals -> region grow -> colours -> PLY out)
against the CPU oracle + glibc rand(), mirroring TMC3.cpp:202-229."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from buildingsegment_amd import synth
from test_host_ply import read_out, write_ply

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_matches_oracle_pipeline(oracle, tmp_path):
    exe = os.path.join(ROOT, "host", "tmc3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    xyz = synth.plane_cube()[:25000].astype(np.int64) + np.array([12345, -777, 50])  # un-shifted input
    src, dst = str(tmp_path / "in.ply"), str(tmp_path / "out.ply")
    metres = (xyz + np.where(xyz >= 0, 0.5, -0.5)) / 1000.0  # trunc(v*1000) recovers xyz
    write_ply(src, metres, np.zeros((len(xyz), 3), np.uint8))
    subprocess.check_call([exe, "-a=" + src, "-s=" + dst])
    _, rec, body = read_out(dst)
    assert body == 27 * len(xyz)
    shifted = (xyz - xyz.min(0)).astype(np.int32)
    assert np.array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), shifted.astype(np.float64))
    neigh, normals = oracle.knn_normals(shifted, k=15)
    _, planes = oracle.region_grow(shifted, normals, neigh)
    libc = ctypes.CDLL(None)
    libc.srand(1)
    want = np.zeros((len(xyz), 3), np.uint8)
    for i in range(len(planes["id"])):
        col = [55 + libc.rand() % 200 for _ in range(3)]
        want[planes["point_idx"][planes["offset"][i]:planes["offset"][i + 1]]] = col
    got = np.stack([rec["g"], rec["b"], rec["r"]], 1)  # file order = internal slots 0,1,2
    assert np.array_equal(got, want)


def test_cli_raster_branch(oracle, tmp_path):
    """--raster=<prefix>: the 2-D branch of the reference's main (TMC3.cpp:223-225, commented
    out there): buildingSeg::compute_gird_picture + save_image.  The three PNGs must decode to
    the pixels save_image derives from the oracle's image."""
    import importlib.util
    from buildingsegment_amd import api
    exe = os.path.join(ROOT, "host", "tmc3")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    xyz = synth.plane_cube()[:20000].astype(np.int64) + np.array([-4321, 99, 7])
    src, dst = str(tmp_path / "in.ply"), str(tmp_path / "out.ply")
    metres = (xyz + np.where(xyz >= 0, 0.5, -0.5)) / 1000.0
    write_ply(src, metres, np.zeros((len(xyz), 3), np.uint8))
    prefix = str(tmp_path / "r_")
    subprocess.check_call([exe, "-a=" + src, "-s=" + dst, "--raster=" + prefix])
    shifted = (xyz - xyz.min(0)).astype(np.int32)
    img, _ = oracle.grid_picture(shifted)
    want = api.save_image(img, str(tmp_path / "want_"))
    spec = importlib.util.spec_from_file_location("mgr", os.path.join(ROOT, "tests", "golden", "make_golden_raster.py"))
    mgr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgr)
    for name, w in zip(("height", "density", "density_height"), want):
        assert np.array_equal(mgr.read_png(prefix + name + ".png"), w), name
