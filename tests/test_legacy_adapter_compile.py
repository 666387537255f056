"""Build-container-only guard of the drop-in boundary: the reference's own call-site lines
(tmc3/TMC3.cpp:210-218, extracted from the reference tree at test time, never stored here)
must compile UNCHANGED against the reference's real PCCPointSet.h with host/bs_legacy.hpp
(-DBS_LEGACY_PCC) providing buildingSeg / get_Normal_and_K_neighbor / seg_plane / plane.
Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/tmc3"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent (GPU box)")
def test_reference_call_sites_compile_against_the_adapter(tmp_path):
    lines = open(os.path.join(REF, "TMC3.cpp"), errors="replace").read().replace("\r", "").split("\n")
    body = "\n".join(lines[209:218])  # TMC3.cpp:210-218: buildingSeg ctor ... set_plane_color
    assert "get_Normal_and_K_neighbor<15>(pointCloud, normal, neigh);" in body and "h.set_plane_color(" in body
    src = tmp_path / "callsite.cpp"
    src.write_text(
        '#include <string>\n#include <vector>\n#include <cmath>\n#include <cstdlib>\n#include <memory>\n'
        '#include "PCCPointSet.h"\n#define BS_LEGACY_PCC\n#include "bs_legacy.hpp"\n'
        "using namespace pcc;\nusing namespace std;\n"
        "int reference_main_body(PCCPointSet3& pointCloud)\n{\n" + body + "\n  return (int)plances.size();\n}\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", REF, "-I", os.path.join(ROOT, "host"),
                           "-I", os.path.join(ROOT, "include"), str(src)])
