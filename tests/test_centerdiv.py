"""bs_centerdiv.h (the grower's cheap exact plane-centre division) against the
reference expression `(int32_t)((uint64_t)(int64_t)c / n)` of
tmc3/my_function.cpp:249-250, compiled for the host: edge values, every small
n, and 20 M random (c, n) pairs.  The device build differs only in using
v_rcp_f64 for the reciprocal estimate, which the integer correction step absorbs
(the GPU parity tests cover that side)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_centerdiv_matches_reference_expression(tmp_path):
    exe = str(tmp_path / "centerdiv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                           "-I", os.path.join(ROOT, "buildingsegment_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "centerdiv_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad=0" in out.stdout
