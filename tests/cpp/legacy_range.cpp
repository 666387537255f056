// The drop-in get_Normal_and_K_neighbor (host/bs_legacy.hpp) on a cloud outside the exact
// domain (|c| >= 2^23 mm, e.g. mm-georeferenced data): the reference would run, this build
// must fail LOUDLY with BS_ERR_RANGE and name the remedy -- and succeed once the cloud is
// shifted the way the buildingSeg constructor does (TMC3.cpp:55-73).
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "bs_pointset.hpp"
#include "bs_legacy.hpp"

int main()
{
  bs::PointSet3 cloud;
  const int side = 40;
  cloud.resize(side * side);
  for (int i = 0; i < side * side; i++) {
    cloud[i][0] = 9000000 + (i % side) * 50;  // 9 km in mm: outside |c| < 2^23
    cloud[i][1] = 9000000 + (i / side) * 50;
    cloud[i][2] = (i * 7) % 3;
  }
  std::vector<bs::Vec3<double>> normal;
  std::vector<std::vector<int>> neigh;
  bool threw = false;
  try {
    bs::get_Normal_and_K_neighbor<15>(cloud, normal, neigh);
  } catch (const std::runtime_error& e) {
    threw = strstr(e.what(), "2^23") != nullptr && strstr(e.what(), "shift the cloud") != nullptr;
    printf("range error: %s\n", e.what());
  }
  if (!threw)
    return 1;
  bs::buildingSeg_t<bs::PointSet3> seg(cloud);  // the reference's pre-processing: shift to the bbox origin
  bs::get_Normal_and_K_neighbor<15>(cloud, normal, neigh);
  if (neigh.size() != (size_t)side * side || neigh[0].size() != 15 || neigh[0][0] != 0)
    return 2;
  printf("ok after shift\n");
  return 0;
}
