// bs_legacy.hpp -- header-only legacy adapter: the reference's C++ call
// signatures for the hot path, implemented on the C ABI (include/bs_api.h).
//
//   struct plane                          /root/reference/tmc3/my_function.h:25-30
//   get_Normal_and_K_neighbor<K>(...)     /root/reference/tmc3/my_function.h:48-85
//   class seg_plane                       /root/reference/tmc3/my_function.h:89-123
//     get_planes()                        /root/reference/tmc3/my_function.cpp:180-217
//     set_plane_color(planes)             /root/reference/tmc3/my_function.cpp:260-275
//   class buildingSeg (2-D raster branch) /root/reference/tmc3/TMC3.cpp:50-200
//     buildingSeg(cloud)                  TMC3.cpp:55-79  (bbox, shift, image dims)
//     compute_gird_picture()              TMC3.cpp:123-174 (+ groundTH, :183-199)
//     save_image(prefix), pixel(x,y,c)    TMC3.cpp:81-121
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
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <vector>

#include "bs_api.h"

namespace bs {

inline bs_ctx* legacy_ctx()
{
  static bs_ctx* ctx = nullptr;
  if (!ctx) {
    // the library writes bs_timings whole into caller memory: a host compiled against another header version must
    // not get that far
    if (bs_api_version() != BS_API_VERSION || bs_sizeof_timings() != (int64_t)sizeof(bs_timings))
      throw std::runtime_error("libbuildingsegment_hip.so is API version " + std::to_string(bs_api_version()) +
                               ", this host was compiled against version " + std::to_string(BS_API_VERSION));
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

// ---- 2-D raster branch ------------------------------------------------------
// 8-bit RGB PNG with stored (uncompressed) deflate blocks: enough for a debug
// image dump, no dependency.  (The reference uses stb_image_write.)
inline bool write_png_rgb8(const std::string& path, int w, int h, const std::vector<uint8_t>& rgb)
{
  auto crc32 = [](const uint8_t* d, size_t n, uint32_t c) {
    c = ~c;
    for (size_t i = 0; i < n; i++) {
      c ^= d[i];
      for (int k = 0; k < 8; k++)
        c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return ~c;
  };
  auto be32 = [](std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8)
      v.push_back((uint8_t)(x >> s));
  };
  std::vector<uint8_t> raw;  // filter byte 0 + row
  raw.reserve((size_t)h * (3 * (size_t)w + 1));
  for (int y = 0; y < h; y++) {
    raw.push_back(0);
    raw.insert(raw.end(), rgb.begin() + (size_t)y * 3 * w, rgb.begin() + (size_t)(y + 1) * 3 * w);
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (size_t pos = 0; pos < raw.size() || pos == 0;) {
    const size_t len = std::min<size_t>(65535, raw.size() - pos);
    const bool last = pos + len >= raw.size();
    z.push_back(last ? 1 : 0);
    z.push_back((uint8_t)(len & 255));
    z.push_back((uint8_t)(len >> 8));
    z.push_back((uint8_t)(~len & 255));
    z.push_back((uint8_t)((~len >> 8) & 255));
    for (size_t i = 0; i < len; i++) {
      a = (a + raw[pos + i]) % 65521u;
      b = (b + a) % 65521u;
    }
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
    pos += len;
    if (last)
      break;
  }
  be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(tag, tag + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size(), 0));
  };
  std::vector<uint8_t> ihdr;
  be32(ihdr, (uint32_t)w);
  be32(ihdr, (uint32_t)h);
  const uint8_t tail[5] = {8, 2, 0, 0, 0};
  ihdr.insert(ihdr.end(), tail, tail + 5);
  chunk("IHDR", ihdr);
  chunk("IDAT", z);
  chunk("IEND", {});
  FILE* f = fopen(path.c_str(), "wb");
  if (!f)
    return false;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}

// The reference's buildingSeg (TMC3.cpp:50-200) on the C ABI.  As in the
// reference, the constructor keeps its own copy of the cloud, shifts BOTH the
// copy and the caller's cloud to the bounding-box origin and sizes the image.
template <class Cloud>
class buildingSeg_t {
public:
  Cloud pointcloud;

  explicit buildingSeg_t(Cloud& cloud) : pointcloud(cloud)
  {
    const size_t n = cloud.getPointCount();
    for (int k = 0; k < 3; k++) {
      box_min[k] = std::numeric_limits<int32_t>::max();
      box_max[k] = std::numeric_limits<int32_t>::lowest();
    }
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        const int32_t v = cloud[i][k];
        if (v > box_max[k])
          box_max[k] = v;
        if (v < box_min[k])
          box_min[k] = v;
      }
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        pointcloud[i][k] -= box_min[k];
        cloud[i][k] = pointcloud[i][k];
      }
    int32_t ext[3] = {box_max[0] - box_min[0], box_max[1] - box_min[1], box_max[2] - box_min[2]};
    if (n == 0 || bs_grid_dims(ext, bin, &width, &height) != BS_OK)
      throw std::runtime_error("buildingSeg: empty cloud");
    image.assign((size_t)width * height * channels, 0.0);
  }

  double& pixel(int x, int y, int channel) { return image[((size_t)y * width + x) * channels + channel]; }

  void compute_gird_picture()
  {
    const int32_t ext[3] = {box_max[0] - box_min[0], box_max[1] - box_min[1], box_max[2] - box_min[2]};
    legacy_check(bs_grid_picture(legacy_ctx(), reinterpret_cast<const int32_t*>(&pointcloud[0]),
                                 (int64_t)pointcloud.getPointCount(), ext, bin, bin_height, image.data(), &ground_th));
  }

  // three PNGs: mean height (red), density (green), third channel (green; never written by
  // compute_gird_picture, i.e. black).  ASCII suffixes replace the reference's Chinese captions.
  void save_image(const std::string& savePath)
  {
    double mx[3] = {0, 0, 0};
    for (int i = 0; i < width; i++)
      for (int j = 0; j < height; j++)
        for (int c = 0; c < channels; c++)
          if (mx[c] < pixel(i, j, c))
            mx[c] = pixel(i, j, c);
    const char* names[3] = {"height.png", "density.png", "density_height.png"};
    const int slot[3] = {0, 1, 1};
    for (int c = 0; c < 3; c++) {
      std::vector<uint8_t> img((size_t)width * height * channels, 0);
      if (mx[c] != 0)
        for (int i = 0; i < width; i++)
          for (int j = 0; j < height; j++)
            img[((size_t)i + (size_t)j * width) * channels + slot[c]] = (uint8_t)(255.0 * (1.0 * pixel(i, j, c) / mx[c]));
      if (!write_png_rgb8(savePath + names[c], width, height, img))
        throw std::runtime_error("save_image: cannot write " + savePath + names[c]);
    }
  }

  int32_t box_min[3], box_max[3];
  int32_t bin = 100, bin_height = 1000;  // TMC3.cpp:179
  int32_t width = 0, height = 0, channels = 3;
  double ground_th = 0;
  std::vector<double> image;
};

}  // namespace bs

#ifdef BS_LEGACY_PCC
using buildingSeg = bs::buildingSeg_t<pcc::PCCPointSet3>;
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
