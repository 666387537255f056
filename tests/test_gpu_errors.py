"""Error behaviour of the boundary on the GPU: 'nothing aborts' (include/bs_api.h)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_region_grow_dev_rejects_out_of_range_neighbours(gpu_ctx, oracle):
    """A foreign neighbour array handed over as a DEVICE pointer is range-checked on the device
    before anything dereferences it (the host entry point checks on the host)."""
    import torch
    from buildingsegment_amd import api, synth
    xyz = synth.plane_cube()[:5000].copy()
    neigh, normals = oracle.knn_normals(xyz, k=15)
    p = api.default_params(k=15)
    dev = torch.device("cuda", 0)
    d_xyz, d_nr = torch.from_numpy(xyz).to(dev), torch.from_numpy(normals).to(dev)
    d_lab = torch.empty(len(xyz), dtype=torch.int32, device=dev)
    for bad_value in (len(xyz), -1, 2**31 - 1):
        bad = neigh.copy()
        bad[1234, 7] = bad_value
        d_ng = torch.from_numpy(bad).to(dev)
        torch.cuda.synchronize()
        with pytest.raises(api.BsError) as e:
            gpu_ctx.region_grow_dev(d_xyz.data_ptr(), d_nr.data_ptr(), d_ng.data_ptr(), len(xyz), d_lab.data_ptr(), p)
        assert e.value.status == -1 and "neighbour index" in str(e.value)
    d_ng = torch.from_numpy(neigh).to(dev)
    torch.cuda.synchronize()
    gpu_ctx.region_grow_dev(d_xyz.data_ptr(), d_nr.data_ptr(), d_ng.data_ptr(), len(xyz), d_lab.data_ptr(), p)
    pi, _ = oracle.region_grow(xyz, normals, neigh)
    assert np.array_equal(d_lab.cpu().numpy(), pi)


def test_legacy_adapter_reports_the_domain_limit(tmp_path):
    exe = str(tmp_path / "legacy_range")
    lib = os.path.join(ROOT, "buildingsegment_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "host"),
                           os.path.join(ROOT, "tests", "cpp", "legacy_range.cpp"), "-o", exe, "-L", lib,
                           "-lbuildingsegment_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok after shift" in out.stdout
