// Driver for the reference's OWN PLY reader / writer (tmc3/ply.cpp, compiled from where it lies
// under /root/reference by oracle/ref/build_ref.sh; this file holds none of its text).
// TEST INFRASTRUCTURE: pins host/bs_ply.cpp byte for byte (tests/golden/make_golden_ply.py).
//
// usage: ref_ply <in.ply> <scale> <out.ply> <ascii 0|1> <dump.bin>
//   ply::read(in, {"x","y","z"}, scale, cloud)              (TMC3.cpp:208)
//   dump: int64 n, int32 has_colors, int32 xyz[n*3], uint16 colors[n*3] (internal G,B,R slots)
//   ply::write(cloud, {"x","y","z"}, 1.0, {0,0,0}, out, ascii)  (TMC3.cpp:221)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "PCCPointSet.h"
#include "ply.h"

int main(int argc, char** argv)
{
  if (argc < 6)
    return 2;
  pcc::PCCPointSet3 cloud;
  if (!pcc::ply::read(argv[1], {"x", "y", "z"}, atof(argv[2]), cloud))
    return 3;
  FILE* f = fopen(argv[5], "wb");
  if (!f)
    return 4;
  const int64_t n = (int64_t)cloud.getPointCount();
  const int32_t hc = cloud.hasColors() ? 1 : 0;
  fwrite(&n, 8, 1, f);
  fwrite(&hc, 4, 1, f);
  for (int64_t i = 0; i < n; i++) {
    int32_t p[3] = {cloud[i][0], cloud[i][1], cloud[i][2]};
    fwrite(p, 4, 3, f);
  }
  for (int64_t i = 0; i < n && hc; i++) {
    uint16_t c[3] = {cloud.getColor(i)[0], cloud.getColor(i)[1], cloud.getColor(i)[2]};
    fwrite(c, 2, 3, f);
  }
  fclose(f);
  if (!pcc::ply::write(cloud, {"x", "y", "z"}, 1.0, {0, 0, 0}, argv[3], atoi(argv[4]) != 0))
    return 5;
  return 0;
}
