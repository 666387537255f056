// bs_raster.hip -- the reference's 2-D branch: density / height
// raster of the shifted cloud, bit-identical to buildingSeg::groundTH +
// compute_gird_picture (/root/reference/tmc3/TMC3.cpp:123-174,183-199).
//
// The reference splats every point at or above the ground threshold bilinearly
// into 4 pixels of a 100-mm grid, accumulating f64 sums IN POINT ORDER; those
// sums are order dependent, so atomics on doubles cannot reproduce them.  The
// device path makes the order explicit instead:
//   1. z histogram (integer atomics, order free) -> ground threshold
//   2. one (pixel, 4*i + corner) pair per contribution, pixel = npix for points
//      below the threshold; per-pixel counts with integer atomics
//   3. stable LSD radix sort of the pairs by pixel (rocPRIM via hipCUB): inside a
//      pixel the contributions stay in point order
//   4. one thread per pixel walks its segment sequentially with the reference's
//      own f64 expressions (no FMA), then the two per-pixel passes (mean height,
//      log density + 20) with the shared deterministic log of bs_detmath.h.
// HBM-bound helper work: 12 B/pt read + 64 B/pt of sort traffic per radix pass.
#include <hipcub/hipcub.hpp>

#include "../../include/bs_detmath.h"
#include "bs_common.h"

namespace bs {
namespace {

// Height histogram.  A city block has a few dozen 1000-mm height bins: global atomics
// on so few addresses serialise in L2 (10 M points: 18 ms), so every block counts in
// LDS first and flushes its non-zero bins once (bins >= ZH_LDS go to HBM directly).
constexpr int ZH_LDS = 4096;
constexpr int ZH_PER_THREAD = 32;

__global__ __launch_bounds__(256) void zhist_kernel(const int32_t* __restrict__ xyz, int64_t n, int3 extent,
                                                    int bin_height, int* __restrict__ hist, int* __restrict__ bad)
{
  __shared__ int lh[ZH_LDS];
  for (int t = threadIdx.x; t < ZH_LDS; t += blockDim.x)
    lh[t] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * blockDim.x * ZH_PER_THREAD;
  for (int r = 0; r < ZH_PER_THREAD; r++) {
    const int64_t i = base + (int64_t)r * blockDim.x + threadIdx.x;
    if (i >= n)
      break;
    const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (x < 0 || y < 0 || z < 0 || x > extent.x || y > extent.y || z > extent.z) {
      *bad = 1;  // not the shifted cloud this extent belongs to: the splat would leave the image
      continue;
    }
    const int b = z / bin_height;
    if (b < ZH_LDS)
      atomicAdd(&lh[b], 1);
    else
      atomicAdd(&hist[b], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ZH_LDS; t += blockDim.x)
    if (lh[t])
      atomicAdd(&hist[t], lh[t]);
}

// groundTH (TMC3.cpp:183-199): first height bin at which the running count exceeds n/2
__global__ void ground_th_kernel(const int* __restrict__ hist, int64_t nb, int64_t n, int bin_height, double* th)
{
  const int TH = (int)(n / 2);
  int total = 0;
  int64_t b;
  for (b = 0; b < nb; b++) {
    total += hist[b];
    if (total > TH)
      break;
  }
  *th = (double)(int)(b * bin_height);
}

__global__ void emit_pairs_kernel(const int32_t* __restrict__ xyz, int64_t n, int bin, int width, uint32_t npix,
                                  const double* __restrict__ th, uint32_t* __restrict__ keys,
                                  uint32_t* __restrict__ vals, int* __restrict__ cnt)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  const bool keep = !((double)pz < *th);  // TMC3.cpp:134
  const int x = px / bin, y = py / bin;
#pragma unroll
  for (int c = 0; c < 4; c++) {  // c = 2*xi + yi: the reference's loop order (TMC3.cpp:131-132)
    const int xi = c >> 1, yi = c & 1;
    const uint32_t pix = (uint32_t)((int64_t)(y + yi) * width + (x + xi));
    keys[4 * i + c] = keep ? pix : npix;
    vals[4 * i + c] = (uint32_t)(4 * i + c);
    if (keep)
      atomicAdd(&cnt[pix], 1);
  }
}

__global__ void accumulate_kernel(const int32_t* __restrict__ xyz, int bin, uint32_t npix,
                                  const int* __restrict__ off, const int* __restrict__ cnt,
                                  const uint32_t* __restrict__ vals, double* __restrict__ image)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix)
    return;
  double c0 = 0.0, c1 = 0.0;
  const int e0 = off[p], e1 = e0 + cnt[p];
  for (int e = e0; e < e1; e++) {
    const uint32_t v = vals[e];
    const int64_t i = v >> 2;
    const int xi = (v >> 1) & 1, yi = v & 1;
    const int px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const int x = px / bin, y = py / bin;
    const double w = 1.0 * px / bin - x;  // TMC3.cpp:136-137
    const double h = 1.0 * py / bin - y;
    const double s = ((xi == 1) ? w : (1 - w)) * ((yi == 1) ? h : (1 - h));
    c1 += s;       // TMC3.cpp:139
    c0 += s * pz;  // TMC3.cpp:140
  }
  if (c1 != 0)  // mean height, TMC3.cpp:148-153
    c0 = c0 / c1;
  c1 = bs_det_log(c1 + 1);  // TMC3.cpp:155-160
  if (c1 != 0)
    c1 += 20;
  image[3 * (int64_t)p] = c0;
  image[3 * (int64_t)p + 1] = c1;
  image[3 * (int64_t)p + 2] = 0.0;
}

inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

}  // namespace
}  // namespace bs

using namespace bs;

extern "C" int bs_grid_dims(const int32_t* extent, int32_t bin, int32_t* width, int32_t* height)
{
  if (!extent || !width || !height || bin <= 0 || extent[0] < 0 || extent[1] < 0)
    return BS_ERR_INVALID;
  *width = extent[0] / bin + 2;   // TMC3.cpp:75
  *height = extent[1] / bin + 2;  // TMC3.cpp:76
  return BS_OK;
}

extern "C" int bs_grid_picture_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, const int32_t* extent, int32_t bin,
                                   int32_t bin_height, double* d_image, double* ground_th)
{
  if (!ctx)
    return BS_ERR_INVALID;
  int32_t width = 0, height = 0;
  if (!d_xyz || !d_image || n <= 0 || bin_height <= 0 || bs_grid_dims(extent, bin, &width, &height) != BS_OK ||
      extent[2] < 0)
    return fail(ctx, BS_ERR_INVALID, "null pointer, empty cloud or bad raster parameters");
  const int64_t npix64 = (int64_t)width * height;
  if (n >= (1ll << 29) || npix64 >= (1ll << 31) - 1)
    return fail(ctx, BS_ERR_RANGE, "raster: more than 2^29 points or 2^31 pixels");
  const uint32_t npix = (uint32_t)npix64;
  const int64_t nb = (int64_t)extent[2] / bin_height + 1;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t m = 4 * n;
  BS_HIP(ctx, ctx->rs_keys_in.reserve(sizeof(uint32_t) * m));
  BS_HIP(ctx, ctx->rs_keys_out.reserve(sizeof(uint32_t) * m));
  BS_HIP(ctx, ctx->rs_vals_in.reserve(sizeof(uint32_t) * m));
  BS_HIP(ctx, ctx->rs_vals_out.reserve(sizeof(uint32_t) * m));
  // cnt[npix] | off[npix] | hist[nb] | bad | pad | th (double, 8-aligned)
  const size_t ints = (size_t)2 * npix + (size_t)nb + 2;
  const size_t th_off = ((ints * sizeof(int) + 7) / 8) * 8;
  BS_HIP(ctx, ctx->rs_cnt.reserve(th_off + sizeof(double)));
  int* cnt = ctx->rs_cnt.as<int>();
  int* off = cnt + npix;
  int* hist = off + npix;
  int* bad = hist + nb;
  double* d_th = reinterpret_cast<double*>(ctx->rs_cnt.as<char>() + th_off);
  BS_HIP(ctx, hipMemsetAsync(cnt, 0, th_off + sizeof(double), st));

  zhist_kernel<<<nblk(n, 256 * ZH_PER_THREAD), 256, 0, st>>>(d_xyz, n, make_int3(extent[0], extent[1], extent[2]), bin_height, hist, bad);
  ground_th_kernel<<<1, 1, 0, st>>>(hist, nb, n, bin_height, d_th);
  uint32_t* keys_in = ctx->rs_keys_in.as<uint32_t>();
  uint32_t* vals_in = ctx->rs_vals_in.as<uint32_t>();
  uint32_t* keys_out = ctx->rs_keys_out.as<uint32_t>();
  uint32_t* vals_out = ctx->rs_vals_out.as<uint32_t>();
  emit_pairs_kernel<<<nblk(n, 256), 256, 0, st>>>(d_xyz, n, bin, width, npix, d_th, keys_in, vals_in, cnt);
  int end_bit = 1;
  while (end_bit < 32 && (1ull << end_bit) <= (unsigned long long)npix)
    end_bit++;
  size_t tmp_sort = 0, tmp_scan = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, keys_in, keys_out, vals_in, vals_out, (int)m, 0,
                                                 end_bit, st));
  BS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, cnt, off, (int)npix, st));
  BS_HIP(ctx, ctx->rs_tmp.reserve(std::max(tmp_sort, tmp_scan)));
  size_t tb = ctx->rs_tmp.cap;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->rs_tmp.p, tb, keys_in, keys_out, vals_in, vals_out, (int)m, 0,
                                                 end_bit, st));
  tb = ctx->rs_tmp.cap;
  BS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->rs_tmp.p, tb, cnt, off, (int)npix, st));
  accumulate_kernel<<<nblk(npix, 64), 64, 0, st>>>(d_xyz, bin, npix, off, cnt, vals_out, d_image);
  int h_bad = 0;
  double h_th = 0;
  BS_HIP(ctx, hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipMemcpyAsync(&h_th, d_th, sizeof(double), hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  if (h_bad)
    return fail(ctx, BS_ERR_RANGE, "raster: a coordinate lies outside [0, extent] (cloud not shifted to its bounding box?)");
  if (ground_th)
    *ground_th = h_th;
  return BS_OK;
}

extern "C" int bs_grid_picture(bs_ctx* ctx, const int32_t* xyz, int64_t n, const int32_t* extent, int32_t bin,
                               int32_t bin_height, double* image, double* ground_th)
{
  if (!ctx)
    return BS_ERR_INVALID;
  int32_t width = 0, height = 0;
  if (!xyz || !image || n <= 0 || bs_grid_dims(extent, bin, &width, &height) != BS_OK)
    return fail(ctx, BS_ERR_INVALID, "null pointer, empty cloud or bad raster parameters");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  const size_t img_bytes = sizeof(double) * 3 * (size_t)width * height;
  BS_HIP(ctx, ctx->d_xyz_h.reserve(sizeof(int32_t) * 3 * n));
  BS_HIP(ctx, ctx->rs_img.reserve(img_bytes));
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_xyz_h.p, xyz, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
  const int rc = bs_grid_picture_dev(ctx, ctx->d_xyz_h.as<int32_t>(), n, extent, bin, bin_height,
                                     ctx->rs_img.as<double>(), ground_th);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipMemcpyAsync(image, ctx->rs_img.p, img_bytes, hipMemcpyDeviceToHost, ctx->stream));
  BS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BS_OK;
}
