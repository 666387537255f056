"""In-tree build of libbuildingsegment_hip.so (hipcc, gfx950 only)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbuildingsegment_hip.so")
SOURCES = ["bs_capi.hip", "bs_grid.hip", "bs_knn.hip", "bs_grow.hip", "bs_grow_spec.hip", "bs_prepost.hip", "bs_raster.hip"]
HEADERS = ["bs_common.h", "bs_normal.h", "bs_centerdiv.h", "../../include/bs_api.h", "../../include/bs_detmath.h"]
# -ffp-contract=off: no FMA fusion anywhere -- host and device must round identically.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
