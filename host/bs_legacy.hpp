// bs_legacy.hpp -- header-only legacy adapter: the reference's C++ call
// signatures for the hot path, implemented on the C ABI (include/bs_api.h).
//
//   struct plane                          /root/reference/tmc3/my_function.h:25-30
//   get_Normal_and_K_neighbor<K>(...)     /root/reference/tmc3/my_function.h:48-85
//   class seg_plane                       /root/reference/tmc3/my_function.h:89-123
//     get_planes()                        /root/reference/tmc3/my_function.cpp:180-217
//     set_plane_color(planes)             /root/reference/tmc3/my_function.cpp:260-275
//
// Generic over the cloud / vector types so that it works both with the
// reference's pcc::PCCPointSet3 + pcc::Vec3<T> (drop-in: define
// BS_LEGACY_PCC after including PCCPointSet.h) and with bs::PointSet3.
// Requirements on Cloud: getPointCount(), operator[](i) -> 3 x int32
// contiguous (&cloud[0] is the AoS base, PCCPointSet.h:271-275,605), public
// std::vector<int> planeIdx, hasColors(), setColor(i, Vec3<uint16_t>).
// Errors (N < K, HIP failure, colourless cloud) throw std::runtime_error where
// the reference has undefined behaviour.  The stray output.ply write of
// my_function.h:81 is intentionally not reproduced.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "bs_api.h"

namespace bs {

inline bs_ctx* legacy_ctx()
{
  static bs_ctx* ctx = nullptr;
  if (!ctx) {
    int rc = bs_create(0, &ctx);
    if (rc != BS_OK)
      throw std::runtime_error(std::string("bs_create: ") + bs_strerror(rc));
  }
  return ctx;
}

inline void legacy_check(int rc)
{
  if (rc != BS_OK)
    throw std::runtime_error(std::string(bs_strerror(rc)) + ": " + bs_last_error(legacy_ctx()));
}

template <class VecD, class VecI>
struct plane_t {
  int id;  // > 0
  VecD normal;
  VecI center;
  std::vector<int> pointIdx;
};

template <int K, class Cloud, class VecD>
void get_Normal_and_K_neighbor(Cloud& pointCloud, std::vector<VecD>& normal, std::vector<std::vector<int>>& neigh)
{
  static_assert(sizeof(pointCloud[0]) == 3 * sizeof(int32_t), "positions must be int32 AoS");
  static_assert(sizeof(VecD) == 3 * sizeof(double), "normals must be f64 AoS");
  const int64_t n = (int64_t)pointCloud.getPointCount();
  normal.resize(n);
  neigh.resize(n);
  if (n == 0)
    return;
  bs_params p;
  bs_params_default(&p);
  p.k = K;
  std::vector<int32_t> flat((size_t)n * K);
  legacy_check(bs_knn_normals(legacy_ctx(), reinterpret_cast<const int32_t*>(&pointCloud[0]), n, &p, flat.data(),
                              reinterpret_cast<double*>(normal.data())));
  for (int64_t i = 0; i < n; i++)  // the reference's vector<vector<int>> surface (my_function.h:49,77)
    neigh[i].assign(flat.begin() + i * K, flat.begin() + (i + 1) * K);
}

template <class Cloud, class VecD, class VecI, class VecC>
class seg_plane_t {
public:
  using plane = plane_t<VecD, VecI>;

  seg_plane_t(Cloud& pointCloud, std::vector<VecD>& normal, std::vector<std::vector<int>>& neigh, int num_neigh)
      : Cloud_(pointCloud), normal_(normal), neigh_(neigh), K(num_neigh)
  {
    Cloud_.planeIdx.resize(Cloud_.getPointCount(), -1);  // my_function.h:103
  }

  std::vector<plane> get_planes()
  {
    const int64_t n = (int64_t)Cloud_.getPointCount();
    std::vector<plane> out;
    if (n == 0)
      return out;
    std::vector<int32_t> flat((size_t)n * K);
    for (int64_t i = 0; i < n; i++) {
      if ((int)neigh_[i].size() < K)
        throw std::runtime_error("seg_plane: neighbour list shorter than K (undefined in the reference)");
      for (int j = 0; j < K; j++)
        flat[(size_t)i * K + j] = neigh_[i][j];
    }
    bs_params p;
    bs_params_default(&p);
    p.k = K;
    p.th_thickness = th_thickness;
    p.th_point_count = th_pointCount;
    bs_planes P;
    static_assert(sizeof(int) == sizeof(int32_t), "planeIdx is int32");
    legacy_check(bs_region_grow(legacy_ctx(), reinterpret_cast<const int32_t*>(&Cloud_[0]),
                                reinterpret_cast<const double*>(normal_.data()), flat.data(), n, &p,
                                reinterpret_cast<int32_t*>(Cloud_.planeIdx.data()), &P));
    out.resize(P.n_planes);
    for (int i = 0; i < P.n_planes; i++) {
      out[i].id = P.id[i];
      for (int a = 0; a < 3; a++) {
        out[i].normal[a] = P.normal[3 * i + a];
        out[i].center[a] = P.center[3 * i + a];
      }
      out[i].pointIdx.assign(P.point_idx + P.offset[i], P.point_idx + P.offset[i + 1]);
    }
    bs_planes_free(&P);
    return out;
  }

  // The sequential recursion has no standalone meaning on the device; kept so
  // that code naming it still links.  Always throws.
  bool Broad(int, int) { throw std::runtime_error("seg_plane::Broad is fused into get_planes()"); }

  void set_plane_color(std::vector<plane>& planes)
  {
    if (!Cloud_.hasColors())
      throw std::runtime_error("set_plane_color: cloud has no colours (out-of-bounds write in the reference)");
    const size_t n = Cloud_.getPointCount();
    for (size_t i = 0; i < n; i++)
      Cloud_.setColor(i, VecC{0, 0, 0});  // my_function.cpp:262-264
    for (plane& p : planes) {
      // evaluation order of the three rand() calls in the reference's braced
      // initialiser is left to right (my_function.cpp:269)
      const int c0 = 55 + rand() % 200;
      const int c1 = 55 + rand() % 200;
      const int c2 = 55 + rand() % 200;
      VecC color{(uint16_t)c0, (uint16_t)c1, (uint16_t)c2};
      for (size_t i = 0; i < p.pointIdx.size(); i++)
        Cloud_.setColor(p.pointIdx[i], color);
    }
  }

private:
  Cloud& Cloud_;
  std::vector<VecD>& normal_;
  std::vector<std::vector<int>>& neigh_;
  int K;
  int th_thickness = 300;   // my_function.h:117
  int th_pointCount = 400;  // my_function.h:118
};

}  // namespace bs

#ifdef BS_LEGACY_PCC
// Drop-in names for the reference tree (include PCCPointSet.h first).
using plane = bs::plane_t<pcc::Vec3<double>, pcc::Vec3<int>>;
using seg_plane = bs::seg_plane_t<pcc::PCCPointSet3, pcc::Vec3<double>, pcc::Vec3<int>, pcc::Vec3<pcc::attr_t>>;
template <int K>
inline void get_Normal_and_K_neighbor(pcc::PCCPointSet3& c, std::vector<pcc::Vec3<double>>& n,
                                      std::vector<std::vector<int>>& g)
{
  bs::get_Normal_and_K_neighbor<K>(c, n, g);
}
#endif
