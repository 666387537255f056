"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/bs_api.h declares; no compute is called without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "bs_api.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bs_[a-z_]+)\s*\(", txt)))


def test_header_and_loader_agree():
    from buildingsegment_amd import _lib
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from buildingsegment_amd import _lib, build
    build.build()
    L = _lib.load()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.bs_api_version() == 5 == _lib.API_VERSION
    assert L.bs_sizeof_timings() == C.sizeof(_lib.Timings)
    assert L.bs_strerror(0) == b"ok" and b"2^23" in L.bs_strerror(-2)


def test_params_default_are_reference_literals():
    from buildingsegment_amd import api
    p = api.default_params()
    assert (p.k, p.max_nn, p.radius, p.th_thickness, p.th_point_count, p.cos_th) == (15, 50, 100.0, 300, 400, 0.88)


def test_no_silent_fallback_without_gpu():
    """Without a GPU the product path must fail loudly (BS_ERR_NO_DEVICE), not
    fall back to any CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from buildingsegment_amd import api
    with pytest.raises(api.BsError) as e:
        api.Context(0)
    assert e.value.status == -5


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "buildingsegment_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                s = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in s.replace("no oracle", ""), f"{f} mentions the oracle"
    for f in os.listdir(os.path.join(ROOT, "host")):
        if f.endswith((".h", ".cpp", ".hpp")):
            assert "oracle" not in open(os.path.join(ROOT, "host", f), errors="ignore").read()
    # developer tools outside tests/ never import it either (checkers that do live in tests/tools/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert "import oracle" not in txt and "from oracle" not in txt, f"tools/{f} uses the oracle"
