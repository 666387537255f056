// bs_pointset.hpp -- minimal host data model used by this repo's own CLI and
// host tests.  Same member surface and memory layout as the parts of the
// reference's PCCPointSet3 / Vec3<T> that the hot path touches
// (/root/reference/tmc3/PCCPointSet.h:60-67,266-293,605-606; PCCMath.h:453):
// contiguous AoS int32 positions, uint16 colours stored G,B,R, public planeIdx.
// When integrating into the reference tree use the reference's own headers
// instead -- host/bs_legacy.hpp is generic over both.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bs {

template <class T>
struct Vec3 {
  T data[3];
  Vec3() = default;
  Vec3(T x, T y, T z) : data{x, y, z} {}
  T& operator[](size_t i) { return data[i]; }
  const T& operator[](size_t i) const { return data[i]; }
};

class PointSet3 {
public:
  std::vector<int> planeIdx;  // PCCPointSet.h:67

  size_t getPointCount() const { return positions_.size(); }
  void resize(size_t n)
  {
    positions_.resize(n);
    if (has_colors_)
      colors_.resize(n);
  }
  Vec3<int32_t>& operator[](size_t i) { return positions_[i]; }
  const Vec3<int32_t>& operator[](size_t i) const { return positions_[i]; }
  bool hasColors() const { return has_colors_; }
  void addColors()
  {
    has_colors_ = true;
    colors_.resize(positions_.size());
  }
  Vec3<uint16_t>& getColor(size_t i) { return colors_[i]; }
  const Vec3<uint16_t>& getColor(size_t i) const { return colors_[i]; }
  void setColor(size_t i, const Vec3<uint16_t>& c) { colors_[i] = c; }

private:
  std::vector<Vec3<int32_t>> positions_;
  std::vector<Vec3<uint16_t>> colors_;  // slots: [0]=green [1]=blue [2]=red (ply.cpp:412-414)
  bool has_colors_ = false;
};

}  // namespace bs
