"""include/bs_detmath.h (the normative acos/cos shared by oracle and device)
against libm: <= 2 ulp on the domains FastEigen3x3 uses."""
import numpy as np


def _ulp(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_det_acos_within_2ulp(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-1, 1, 20000), [-1.0, 1.0, 0.0, 0.5, -0.5, 1e-20, -1e-20],
                         1 - np.logspace(-16, -1, 200), -1 + np.logspace(-16, -1, 200)])
    got = np.array([oracle.det_acos(x) for x in xs])
    assert _ulp(got, np.arccos(xs)).max() <= 2.0


def test_det_cos_within_2ulp(oracle):
    rng = np.random.default_rng(1)
    a = rng.uniform(0, np.pi / 3, 20000)       # angle
    b = a + 2.09439510239319549                # angle + 2pi/3
    for xs in (a, b):
        got = np.array([oracle.det_cos(x) for x in xs])
        assert _ulp(got, np.cos(xs)).max() <= 2.0


def test_det_log_within_1ulp(oracle):
    """bs_det_log (density channel of the 2-D raster, TMC3.cpp:158) against libm."""
    rng = np.random.default_rng(2)
    xs = np.concatenate([1 + rng.uniform(0, 10, 20000), 1 + rng.uniform(0, 1e6, 20000), 1 + np.logspace(-16, 3, 500),
                         [1.0, 2.0, np.e, 1.5, 1 + 2.0 ** -30, 4.0, 1e300, 2.0 ** -1000]])
    got = np.array([oracle.det_log(x) for x in xs])
    ref = np.log(xs)
    assert got[0] == got[0] and oracle.det_log(1.0) == 0.0
    nz = ref != 0
    assert _ulp(got[nz], ref[nz]).max() <= 1.0
