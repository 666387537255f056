// bs_ply.hpp -- bulk PLY reader/writer with the reference's file semantics
// (/root/reference/tmc3/ply.h:69-85, ply.cpp:88-186,190-504; SURVEY.md
// Appendix C), written from scratch around whole-block reads instead of the
// reference's six ifstream::read calls per point.
//   read : ascii | binary_little_endian 1.0; positions = (int32) trunc(value * scale);
//          red/green/blue uchar -> colour slots [0]=green [1]=blue [2]=red;
//          unknown scalar properties skipped; face element ignored.
//   write: header text identical to ply.cpp:103-139; binary body 3 x f64 + 3 x u8
//          (green, blue, red) = 27 B per point; ascii uses fixed setprecision(5).
#pragma once
#include <string>

#include "bs_pointset.hpp"

namespace bs {
namespace ply {

bool read(const std::string& file, double positionScale, PointSet3& cloud, std::string* err = nullptr);
bool write(const PointSet3& cloud, double positionScale, const double positionOffset[3], const std::string& file,
           bool asAscii, std::string* err = nullptr);

}  // namespace ply
}  // namespace bs
