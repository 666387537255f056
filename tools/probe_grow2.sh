#!/bin/bash
# tools/probe_grow2.sh <out.so>: the HIP library with cycle stamps (clock64) between the phases of one step of
# grow_spec2_kernel (-DBS_PROBE).  Run with BS_DEBUG=1 BS_LIB_PATH=<out.so>; the host prints the averages per round.
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/buildingsegment_amd/csrc
mkdir -p "$(dirname "$1")"
exec /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -DBS_PROBE ${BS_EXTRA_FLAGS} \
  $C/bs_capi.hip $C/bs_grid.hip $C/bs_knn.hip $C/bs_grow.hip $C/bs_grow_spec.hip $C/bs_prepost.hip $C/bs_raster.hip $C/bs_shard.hip $C/bs_sharded.hip -ldl -o "$1"
