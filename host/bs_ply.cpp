// bs_ply.cpp -- see bs_ply.hpp.
#include "bs_ply.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <vector>

namespace bs {
namespace ply {

namespace {

struct Prop {
  std::string name;
  int size;      // bytes
  char kind;     // 'f' float, 'i' signed, 'u' unsigned
};

bool parse_type(const std::string& t, Prop& p)
{
  if (t == "float64" || t == "double") { p.size = 8; p.kind = 'f'; }
  else if (t == "float" || t == "float32") { p.size = 4; p.kind = 'f'; }
  else if (t == "uint64") { p.size = 8; p.kind = 'u'; }
  else if (t == "int64") { p.size = 8; p.kind = 'i'; }
  else if (t == "uint32" || t == "uint") { p.size = 4; p.kind = 'u'; }
  else if (t == "int32" || t == "int") { p.size = 4; p.kind = 'i'; }
  else if (t == "uint16" || t == "ushort") { p.size = 2; p.kind = 'u'; }
  else if (t == "int16" || t == "short") { p.size = 2; p.kind = 'i'; }
  else if (t == "uint8" || t == "uchar") { p.size = 1; p.kind = 'u'; }
  else if (t == "int8" || t == "char") { p.size = 1; p.kind = 'i'; }
  else return false;
  return true;
}

double load_scalar(const unsigned char* b, const Prop& p)
{
  switch (p.kind) {
  case 'f':
    if (p.size == 8) { double v; memcpy(&v, b, 8); return v; }
    { float v; memcpy(&v, b, 4); return v; }
  case 'i':
    if (p.size == 8) { int64_t v; memcpy(&v, b, 8); return (double)v; }
    if (p.size == 4) { int32_t v; memcpy(&v, b, 4); return v; }
    if (p.size == 2) { int16_t v; memcpy(&v, b, 2); return v; }
    return (int8_t)b[0];
  default:
    if (p.size == 8) { uint64_t v; memcpy(&v, b, 8); return (double)v; }
    if (p.size == 4) { uint32_t v; memcpy(&v, b, 4); return v; }
    if (p.size == 2) { uint16_t v; memcpy(&v, b, 2); return v; }
    return b[0];
  }
}

bool fail(std::string* err, const std::string& m)
{
  if (err)
    *err = m;
  return false;
}

}  // namespace

bool read(const std::string& file, double scale, PointSet3& cloud, std::string* err)
{
  FILE* f = fopen(file.c_str(), "rb");
  if (!f)
    return fail(err, "cannot open " + file);
  std::vector<Prop> props;
  bool ascii = false, in_vertex = false, have_format = false;
  long long count = -1;
  char line[1024];
  bool ok = false;
  if (!fgets(line, sizeof line, f) || strncmp(line, "ply", 3) != 0) {
    fclose(f);
    return fail(err, "not a PLY file");
  }
  while (fgets(line, sizeof line, f)) {
    std::istringstream ss(line);  // separators: space, tab, CR (ply.cpp:227-309)
    std::string tok;
    ss >> tok;
    if (tok == "format") {
      std::string fmt, ver;
      ss >> fmt >> ver;
      if (ver != "1.0") { fclose(f); return fail(err, "unsupported PLY version"); }
      if (fmt == "ascii") ascii = true;
      else if (fmt == "binary_little_endian") ascii = false;
      else { fclose(f); return fail(err, "unsupported PLY format " + fmt); }
      have_format = true;
    } else if (tok == "element") {
      std::string name;
      ss >> name;
      in_vertex = (name == "vertex");
      if (in_vertex)
        ss >> count;
    } else if (tok == "property" && in_vertex) {
      std::string type, name;
      ss >> type;
      if (type == "list") { fclose(f); return fail(err, "list property in vertex element"); }
      ss >> name;
      Prop p;
      p.name = name;
      if (!parse_type(type, p)) { fclose(f); return fail(err, "unknown property type " + type); }
      props.push_back(p);
    } else if (tok == "end_header") {
      ok = true;
      break;
    }
  }
  if (!ok || !have_format || count < 0) {
    fclose(f);
    return fail(err, "malformed PLY header");
  }
  int ix[3] = {-1, -1, -1}, ic[3] = {-1, -1, -1};  // colour slots: green, blue, red
  size_t stride = 0;
  std::vector<size_t> off(props.size());
  for (size_t i = 0; i < props.size(); i++) {
    off[i] = stride;
    stride += props[i].size;
    if (props[i].name == "x") ix[0] = (int)i;
    if (props[i].name == "y") ix[1] = (int)i;
    if (props[i].name == "z") ix[2] = (int)i;
    if (props[i].name == "green" && props[i].size == 1) ic[0] = (int)i;
    if (props[i].name == "blue" && props[i].size == 1) ic[1] = (int)i;
    if (props[i].name == "red" && props[i].size == 1) ic[2] = (int)i;
  }
  if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0) {
    fclose(f);
    return fail(err, "missing x/y/z");
  }
  for (int a = 0; a < 3; a++)
    if (props[ix[a]].size != 4 && props[ix[a]].size != 8) {  // ply.cpp:330-341
      fclose(f);
      return fail(err, "x/y/z must be 4 or 8 bytes");
    }
  const bool colors = ic[0] >= 0 && ic[1] >= 0 && ic[2] >= 0;
  cloud = PointSet3();
  if (colors)
    cloud.addColors();
  cloud.resize((size_t)count);
  if (ascii) {
    std::vector<double> v(props.size());
    for (long long i = 0; i < count; i++) {
      for (size_t k = 0; k < props.size(); k++)
        if (fscanf(f, "%lf", &v[k]) != 1) {
          fclose(f);
          return fail(err, "truncated ascii body");
        }
      for (int a = 0; a < 3; a++)
        cloud[i][a] = (int32_t)(v[ix[a]] * scale);
      if (colors)
        cloud.setColor(i, Vec3<uint16_t>((uint16_t)v[ic[0]], (uint16_t)v[ic[1]], (uint16_t)v[ic[2]]));
    }
  } else {
    const size_t chunk = 1 << 16;
    std::vector<unsigned char> buf(chunk * stride);
    long long done = 0;
    while (done < count) {
      size_t want = (size_t)std::min<long long>(chunk, count - done);
      if (fread(buf.data(), stride, want, f) != want) {
        fclose(f);
        return fail(err, "truncated binary body");
      }
      for (size_t r = 0; r < want; r++) {
        const unsigned char* row = buf.data() + r * stride;
        for (int a = 0; a < 3; a++)
          cloud[done + r][a] = (int32_t)(load_scalar(row + off[ix[a]], props[ix[a]]) * scale);  // ply.cpp:436-465
        if (colors)
          cloud.setColor(done + r, Vec3<uint16_t>(row[off[ic[0]]], row[off[ic[1]]], row[off[ic[2]]]));
      }
      done += (long long)want;
    }
  }
  fclose(f);
  return true;
}

bool write(const PointSet3& cloud, double scale, const double offset[3], const std::string& file, bool asAscii,
           std::string* err)
{
  FILE* f = fopen(file.c_str(), "wb");
  if (!f)
    return fail(err, "cannot open " + file);
  const size_t n = cloud.getPointCount();
  fprintf(f, "ply\n");
  fprintf(f, asAscii ? "format ascii 1.0\n" : "format binary_little_endian 1.0\n");
  fprintf(f, "element vertex %zu\n", n);
  const char* ft = asAscii ? "float" : "float64";
  fprintf(f, "property %s x\nproperty %s y\nproperty %s z\n", ft, ft, ft);
  if (cloud.hasColors())
    fprintf(f, "property uchar green\nproperty uchar blue\nproperty uchar red\n");
  fprintf(f, "element face 0\nproperty list uint8 int32 vertex_index\nend_header\n");
  if (asAscii) {
    for (size_t i = 0; i < n; i++) {
      fprintf(f, "%.5f %.5f %.5f", cloud[i][0] * scale + offset[0], cloud[i][1] * scale + offset[1],
              cloud[i][2] * scale + offset[2]);
      if (cloud.hasColors())
        fprintf(f, " %d %d %d", (int)cloud.getColor(i)[0], (int)cloud.getColor(i)[1], (int)cloud.getColor(i)[2]);
      fprintf(f, "\n");
    }
  } else {
    const size_t stride = 24 + (cloud.hasColors() ? 3 : 0);
    const size_t chunk = 1 << 16;
    std::vector<unsigned char> buf(chunk * stride);
    for (size_t base = 0; base < n; base += chunk) {
      size_t m = std::min(chunk, n - base);
      for (size_t r = 0; r < m; r++) {
        unsigned char* row = buf.data() + r * stride;
        for (int a = 0; a < 3; a++) {
          double v = cloud[base + r][a] * scale + offset[a];
          memcpy(row + 8 * a, &v, 8);
        }
        if (cloud.hasColors())
          for (int a = 0; a < 3; a++)
            row[24 + a] = (uint8_t)cloud.getColor(base + r)[a];  // ply.cpp:170
      }
      if (fwrite(buf.data(), stride, m, f) != m) {
        fclose(f);
        return fail(err, "short write");
      }
    }
  }
  fclose(f);
  return true;
}

}  // namespace ply
}  // namespace bs
