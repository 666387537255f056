// Driver for the VERBATIM reference raster code (struct Box, class buildingSeg:
// constructor shift, groundTH, compute_gird_picture; TMC3.cpp:44-200).  This
// file contains none of the reference's source: build_ref.sh prepends the class
// text, extracted at build time from /root/reference/tmc3/TMC3.cpp, in a
// temporary directory (compiled with -Dprivate=public so that the driver can
// read the image), and only the binary is kept (oracle/_ref/ref_raster).
// TEST INFRASTRUCTURE.
//
// usage: ref_raster <in.bin> <out.bin> [png_prefix]
//   in : int64 n, int32 xyz[n*3]                  (unshifted: the ctor shifts)
//   out: int32 width, int32 height, int32 min[3], int32 max[3], f64 ground_th,
//        f64 image[height*width*3]                 (after compute_gird_picture)
//   png_prefix: also run save_image(prefix) (three PNG files)
#include <cstdint>
#include <cstdio>

int main(int argc, char** argv)
{
  if (argc < 3)
    return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f)
    return 1;
  int64_t n;
  if (fread(&n, 8, 1, f) != 1)
    return 1;
  std::vector<int32_t> xyz(n * 3);
  if (fread(xyz.data(), 4, n * 3, f) != (size_t)(n * 3))
    return 1;
  fclose(f);
  PCCPointSet3 pc;
  pc.resize(n);
  for (int64_t i = 0; i < n; i++)
    pc[i] = Vec3<int32_t>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  buildingSeg seg(pc);
  const double th = seg.groundTH();
  seg.compute_gird_picture();
  FILE* o = fopen(argv[2], "wb");
  if (!o)
    return 1;
  int32_t w = seg.width, h = seg.height;
  fwrite(&w, 4, 1, o);
  fwrite(&h, 4, 1, o);
  for (int k = 0; k < 3; k++) {
    int32_t v = seg.box.min[k];
    fwrite(&v, 4, 1, o);
  }
  for (int k = 0; k < 3; k++) {
    int32_t v = seg.box.max[k];
    fwrite(&v, 4, 1, o);
  }
  fwrite(&th, 8, 1, o);
  fwrite(seg.image.data(), 8, seg.image.size(), o);
  fclose(o);
  if (argc > 3)
    seg.save_image(argv[3]);
  return 0;
}
