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
