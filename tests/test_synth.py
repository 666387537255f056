import numpy as np

from buildingsegment_amd import synth


def test_generators_are_deterministic_and_shaped():
    a, b = synth.plane_cube(), synth.plane_cube()
    assert a.shape == (99946, 3) and a.dtype == np.int32 and np.array_equal(a, b)
    assert a.min() == 0 and a.max() < (1 << 23)
    f = synth.facade(n_side=200)
    assert f.shape == (40000, 3) and f.min(0).tolist() == [0, 0, 0]
    u = synth.urban(123_457, seed=3)
    assert u.shape == (123_457, 3)
    assert not np.array_equal(synth.urban(5000, seed=3), synth.urban(5000, seed=4))
    uni = synth.uniform(10000)
    assert uni.max() < round(50 * 10000 ** (1 / 3))


def test_splitmix64_known_answers():
    # splitmix64 with seed 0: first outputs of the canonical generator
    z = synth.splitmix64(0, 0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_shift_to_origin_mirrors_buildingseg_ctor():
    x = np.array([[5, -3, 7], [9, 0, 7], [6, 2, 11]], np.int32)
    s = synth.shift_to_origin(x)
    assert s.min(0).tolist() == [0, 0, 0] and np.array_equal(s - s[0], x - x[0])
