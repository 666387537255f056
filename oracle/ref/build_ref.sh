#!/bin/sh
# Builds oracle/_ref/ref_stage3 and oracle/_ref/ref_raster from the reference's own sources where
# they lie under /root/reference (SURVEY.md section 8c recipe).  The extracted
# text only ever exists in a temporary directory; nothing but the binary is
# written into the repo (oracle/_ref/ is git-ignored).
set -e
REF=/root/reference/tmc3
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/../_ref"
[ -d "$REF" ] || { echo "no reference tree: skipping oracle/_ref"; exit 0; }
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
{
  printf '#include <string>\n#include <vector>\n#include <cmath>\n#include <cstdlib>\n#include <memory>\n#include "ply.h"\nusing namespace pcc;\nusing namespace std;\n'
  sed -n '25,30p;89,123p' "$REF/my_function.h" | tr -d '\r'
  sed -n '180,275p' "$REF/my_function.cpp" | tr -d '\r'
  cat "$HERE/ref_stage3_driver.cpp"
} > "$TMP/ref_stage3.cpp"
g++ -std=c++17 -O2 -DNDEBUG -w -I"$REF" "$TMP/ref_stage3.cpp" -o "$OUT/ref_stage3" -lpthread
echo "built $OUT/ref_stage3"
# the 2-D raster branch (struct Box + class buildingSeg, TMC3.cpp:44-200); a `#define private public` after the system headers
# lets the driver read the image; stb_image_write.h is the reference's own vendored header
{
  printf '#include <string>\n#include <vector>\n#include <cmath>\n#include <cstdlib>\n#include <limits>\n#include <memory>\n#include "PCCPointSet.h"\n#define STB_IMAGE_WRITE_IMPLEMENTATION\n#include "stb_image_write.h"\nusing namespace pcc;\nusing namespace std;\n#define private public\n'
  sed -n '44,47p;50,200p' "$REF/TMC3.cpp" | tr -d '\r'
  cat "$HERE/ref_raster_driver.cpp"
} > "$TMP/ref_raster.cpp"
g++ -std=c++17 -O2 -DNDEBUG -w -I"$REF" "$TMP/ref_raster.cpp" -o "$OUT/ref_raster"
echo "built $OUT/ref_raster"
# the PLY reader / writer: the reference's ply.cpp compiled where it lies, plus a driver
g++ -std=c++17 -O2 -DNDEBUG -w -I"$REF" "$REF/ply.cpp" "$HERE/ref_ply_driver.cpp" -o "$OUT/ref_ply"
echo "built $OUT/ref_ply"
