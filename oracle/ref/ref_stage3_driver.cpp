// Driver for the VERBATIM reference stage-3 code (seg_plane::get_planes /
// Broad / set_plane_color).  This file contains none of the reference's
// source: build_ref.sh prepends the class text, extracted at build time from
// /root/reference/tmc3/my_function.{h,cpp}, in a temporary directory and only
// the resulting binary is kept (oracle/_ref/ref_stage3).  TEST INFRASTRUCTURE.
//
// usage: ref_stage3 <in.bin> <out.bin>
//   in : int64 n, int32 k, int32 xyz[n*3], f64 normals[n*3], int32 neigh[n*k]
//   out: int32 planeIdx[n], int32 nplanes, then per plane
//        int32 id, f64 normal[3], int32 center[3], int64 size, int32 pointIdx[size]
//        then uint16 colors[n*3] (internal G,B,R slots after set_plane_color)
#include <pthread.h>
#include <cstdio>
#include <cstdint>

struct job {
  const char* in;
  const char* out;
  int rc;
};

static void* run(void* arg)
{
  job* j = (job*)arg;
  j->rc = 1;
  FILE* f = fopen(j->in, "rb");
  if (!f)
    return nullptr;
  int64_t n;
  int32_t k;
  if (fread(&n, 8, 1, f) != 1 || fread(&k, 4, 1, f) != 1)
    return nullptr;
  std::vector<int32_t> xyz(n * 3);
  std::vector<double> nr(n * 3);
  std::vector<int32_t> ng(n * (int64_t)k);
  if (fread(xyz.data(), 4, n * 3, f) != (size_t)(n * 3) || fread(nr.data(), 8, n * 3, f) != (size_t)(n * 3) ||
      fread(ng.data(), 4, n * k, f) != (size_t)(n * k))
    return nullptr;
  fclose(f);

  PCCPointSet3 pc;
  pc.addColors();
  pc.resize(n);
  std::vector<Vec3<double>> normal(n);
  std::vector<std::vector<int>> neigh(n);
  for (int64_t i = 0; i < n; i++) {
    pc[i] = Vec3<int32_t>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    normal[i] = Vec3<double>(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2]);
    neigh[i].assign(ng.begin() + i * k, ng.begin() + (i + 1) * k);
  }
  seg_plane h(pc, normal, neigh, k);
  std::vector<plane> planes = h.get_planes();
  h.set_plane_color(planes);

  FILE* o = fopen(j->out, "wb");
  if (!o)
    return nullptr;
  fwrite(pc.planeIdx.data(), 4, n, o);
  int32_t np = (int32_t)planes.size();
  fwrite(&np, 4, 1, o);
  for (auto& p : planes) {
    int32_t id = p.id;
    fwrite(&id, 4, 1, o);
    double nn[3] = {p.normal[0], p.normal[1], p.normal[2]};
    fwrite(nn, 8, 3, o);
    int32_t cc[3] = {p.center[0], p.center[1], p.center[2]};
    fwrite(cc, 4, 3, o);
    int64_t sz = (int64_t)p.pointIdx.size();
    fwrite(&sz, 8, 1, o);
    fwrite(p.pointIdx.data(), 4, sz, o);
  }
  for (int64_t i = 0; i < n; i++) {
    auto c = pc.getColor(i);
    uint16_t cc[3] = {c[0], c[1], c[2]};
    fwrite(cc, 2, 3, o);
  }
  fclose(o);
  j->rc = 0;
  return nullptr;
}

int main(int argc, char** argv)
{
  if (argc != 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  job j{argv[1], argv[2], 1};
  // the reference recurses once per accepted point (~208 B/frame): give the
  // worker a 16 GiB lazily-committed stack
  pthread_attr_t at;
  pthread_attr_init(&at);
  pthread_attr_setstacksize(&at, (size_t)16 << 30);
  pthread_t t;
  if (pthread_create(&t, &at, run, &j) != 0) {
    fprintf(stderr, "pthread_create failed\n");
    return 3;
  }
  pthread_join(t, nullptr);
  return j.rc;
}
