"""In-tree build of libbuildingsegment_hip.so (hipcc, gfx950 only).

Every translation unit is compiled to its own object (in parallel, only when it or a header is newer) and the
objects are linked into the shared library that travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libbuildingsegment_hip.so")
SOURCES = ["bs_capi.hip", "bs_grid.hip", "bs_knn.hip", "bs_grow.hip", "bs_grow_spec.hip", "bs_prepost.hip", "bs_raster.hip",
           "bs_shard.hip", "bs_sharded.hip"]
HEADERS = ["bs_common.h", "bs_normal.h", "bs_centerdiv.h", "bs_comm.h", "../../include/bs_api.h", "../../include/bs_detmath.h"]
# -ffp-contract=off: no FMA fusion anywhere -- host and device must round identically.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result"]
# (RCCL is NOT linked: bs_sharded.hip resolves librccl.so.1 at run time, so that the one copy already in the
# process -- torch's, under bench.py -- is the one used, and the library loads where no RCCL is installed)
LINK = ["-shared", "-ldl"]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _hdr_mtime() -> float:
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale_objs():
    ht = _hdr_mtime()
    out = []
    for s in _sources():
        o = _obj(s)
        if not os.path.exists(o) or os.path.getmtime(o) < max(ht, os.path.getmtime(os.path.join(CSRC, s))):
            out.append(s)
    return out


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    todo = _sources() if force else _stale_objs()
    objs = [_obj(s) for s in _sources()]
    if not todo and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(o) for o in objs):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def cc(src):
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(cc, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC"] + objs + LINK + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
