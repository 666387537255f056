"""2-D raster on the device (bs_grid_picture / bs_grid_picture_dev through the C
ABI) against the CPU oracle (bit-exact in all three channels: both sides use the
shared deterministic log) and against the golden vectors of the reference's own
buildingSeg code (height channel and density sums exact, log within 1 ulp)."""
import glob
import os

import numpy as np
import pytest

from buildingsegment_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "raster_*.npz")))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_device_vs_reference_golden(gpu_ctx, oracle, path):
    g = np.load(path)
    sh = (g["xyz"] - g["box_min"]).astype(np.int32)
    ext = (g["box_max"] - g["box_min"]).astype(np.int32)
    assert api.grid_dims(ext) == (int(g["width"]), int(g["height"]))
    img, th = gpu_ctx.grid_picture(sh, extent=ext)
    assert th == float(g["ground_th"])
    assert np.array_equal(img[..., 0], g["image"][..., 0])  # mean height: every bit
    assert np.array_equal(img[..., 2], g["image"][..., 2])
    ref1 = g["image"][..., 1]
    assert (np.abs(img[..., 1] - ref1) <= np.spacing(np.abs(ref1))).all()  # log: <= 1 ulp of libm
    oimg, oth = oracle.grid_picture(sh, extent=ext)
    assert oth == th and np.array_equal(img, oimg)  # oracle with the same log: bit-exact
    outs = api.save_image(img, os.path.join(os.environ.get("TMPDIR", "/tmp"), "bs_raster_test_"))
    assert np.array_equal(outs[0], g["png_height"])
    assert (np.abs(outs[1].astype(int) - g["png_density"].astype(int)) <= 1).all()


@pytest.mark.parametrize("n,bin_,bh", [(400_000, 100, 1000), (400_000, 37, 250), (150_000, 1000, 5000)])
def test_device_vs_oracle_urban(gpu_ctx, oracle, n, bin_, bh):
    xyz = synth.shift_to_origin(synth.urban(n, seed=11))
    img, th = gpu_ctx.grid_picture(xyz, bin=bin_, bin_height=bh)
    oimg, oth = oracle.grid_picture(xyz, bin=bin_, bin_height=bh)
    assert th == oth and np.array_equal(img, oimg)
    assert (img[..., 1] != 0).sum() > 20


def test_device_order_dependence_is_reproduced(gpu_ctx, oracle):
    """The f64 sums depend on the point order: a permuted cloud gives (slightly) different
    sums in the reference, and the device must follow the order it is given."""
    rng = np.random.default_rng(3)
    xyz = rng.integers(0, 3000, (200_000, 3)).astype(np.int32)
    xyz[0] = 0
    xyz[1] = 2999
    perm = rng.permutation(len(xyz))
    a, _ = gpu_ctx.grid_picture(xyz)
    b, _ = gpu_ctx.grid_picture(xyz[perm])
    oa, _ = oracle.grid_picture(xyz)
    ob, _ = oracle.grid_picture(xyz[perm])
    assert np.array_equal(a, oa) and np.array_equal(b, ob)
    assert not np.array_equal(oa, ob)  # the test is meaningful: the order matters
    assert np.allclose(a, b, rtol=1e-12, atol=0)


def test_device_resident_variant(gpu_ctx, oracle):
    import torch
    xyz = synth.shift_to_origin(synth.plane_cube())
    ext = xyz.max(0).astype(np.int32)
    w, h = api.grid_dims(ext)
    d_xyz = torch.from_numpy(xyz).cuda()
    d_img = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    th = gpu_ctx.grid_picture_dev(d_xyz.data_ptr(), len(xyz), ext, d_img.data_ptr())
    oimg, oth = oracle.grid_picture(xyz, extent=ext)
    assert th == oth and np.array_equal(d_img.cpu().numpy(), oimg)


def test_errors(gpu_ctx):
    xyz = synth.shift_to_origin(synth.plane_cube()[:1000])
    with pytest.raises(api.BsError):  # not shifted to the origin of this extent
        gpu_ctx.grid_picture(xyz - 5, extent=xyz.max(0))
    with pytest.raises(api.BsError):  # a point beyond the extent
        gpu_ctx.grid_picture(xyz, extent=xyz.max(0) - 1)
    with pytest.raises(ValueError):
        api.grid_dims([10, 10, 10], bin=0)
