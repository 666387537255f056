#!/bin/sh
# Builds oracle/_ref/ref_stage3 from the reference's own stage-3 sources where
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
