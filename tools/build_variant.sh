#!/bin/bash
# tools/build_variant.sh <grow_spec variant .hip> <out .so>: the HIP library with one source swapped (A/B experiments)
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/buildingsegment_amd/csrc
exec /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math ${BS_EXTRA_FLAGS} \
  -I$R/include -I$C $C/bs_capi.hip $C/bs_grid.hip $C/bs_knn.hip $C/bs_grow.hip "$1" $C/bs_prepost.hip $C/bs_raster.hip $C/bs_shard.hip $C/bs_sharded.hip -ldl -o "$2"
