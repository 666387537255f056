// tmc3_main.cpp -- CLI with the reference's surface (tmc3 <x>=<in.ply> <y>=<out.ply>,
// /root/reference/tmc3/TMC3.cpp:202-229, my_function.cpp:147-178) running the
// hot path on the GPU through the legacy adapter.  Flow kept as in main():
// read (scale 1000, trunc) -> bbox shift to >= 0 (buildingSeg ctor,
// TMC3.cpp:55-73) -> get_Normal_and_K_neighbor<15> -> seg_plane::get_planes ->
// set_plane_color -> write (scale 1.0, zero offset, binary).
#include <cstdio>
#include <limits>
#include <string>

#include "bs_legacy.hpp"
#include "bs_ply.hpp"

using Cloud = bs::PointSet3;
using VecD = bs::Vec3<double>;
using VecI = bs::Vec3<int>;
using VecC = bs::Vec3<uint16_t>;

static std::string after_equals(const char* arg)  // Split(path, "=")[1], my_function.cpp:167-172
{
  std::string s(arg);
  size_t p = s.find('=');
  return p == std::string::npos ? s : s.substr(p + 1);
}

int main(int argc, char* argv[])
{
  if (argc < 3) {
    fprintf(stderr, "usage: %s <x>=<in.ply> <y>=<out.ply> [--raster=<png prefix>]\n", argv[0]);
    return 2;
  }
  const std::string in = after_equals(argv[1]), out = after_equals(argv[2]);
  Cloud cloud;
  std::string err;
  if (!bs::ply::read(in, 1000.0, cloud, &err)) {
    fprintf(stderr, "ply read failed: %s\n", err.c_str());
    return 1;
  }
  if (argc > 3 && std::string(argv[3]) == "--io-only") {  // I/O surface check, no GPU involved
    const double z[3] = {0, 0, 0};
    const bool ascii = argc > 4 && std::string(argv[4]) == "ascii";
    return bs::ply::write(cloud, 1.0, z, out, ascii, &err) ? 0 : 1;
  }
  const size_t n = cloud.getPointCount();
  std::string raster_prefix;  // --raster=<prefix>: the 2-D branch the reference's main keeps commented out (TMC3.cpp:223-225)
  for (int a = 3; a < argc; a++)
    if (std::string(argv[a]).rfind("--raster=", 0) == 0)
      raster_prefix = after_equals(argv[a]);
  try {
    bs::buildingSeg_t<Cloud> seg(cloud);  // bounding box + shift to the origin (TMC3.cpp:209)
    std::vector<VecD> normal;
    std::vector<std::vector<int>> neigh;
    bs::get_Normal_and_K_neighbor<15>(cloud, normal, neigh);
    bs::seg_plane_t<Cloud, VecD, VecI, VecC> h(cloud, normal, neigh, 15);
    auto planes = h.get_planes();
    // The reference never seeds rand() (my_function.cpp:269), i.e. it draws from
    // glibc's seed-1 stream; the HIP runtime consumes rand() values during
    // initialisation, so restore that stream before drawing the colours.
    srand(1);
    h.set_plane_color(planes);
    fprintf(stderr, "tmc3: %zu points, %zu planes\n", n, planes.size());
    if (!raster_prefix.empty()) {
      seg.compute_gird_picture();
      seg.save_image(raster_prefix);
      fprintf(stderr, "tmc3: raster %d x %d, ground threshold %.0f mm\n", seg.width, seg.height, seg.ground_th);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "tmc3: %s\n", e.what());
    return 1;
  }
  const double zero[3] = {0, 0, 0};
  if (!bs::ply::write(cloud, 1.0, zero, out, false, &err)) {
    fprintf(stderr, "ply write failed: %s\n", err.c_str());
    return 1;
  }
  return 0;
}
