// bs_prepost.hip -- the O(N) passes either side of the hot path, on the device
// (SURVEY.md 8f-2,3): bounding-box shift of the buildingSeg constructor
// (/root/reference/tmc3/TMC3.cpp:55-73) and the label -> colour scatter of
// seg_plane::set_plane_color (/root/reference/tmc3/my_function.cpp:260-275).
#include <climits>

#include "bs_centerdiv.h"
#include "bs_common.h"

namespace bs {
namespace {

__global__ void minmax_kernel(const int32_t* __restrict__ xyz, int64_t n, int32_t* __restrict__ mn)
{
  int m[3] = {INT_MAX, INT_MAX, INT_MAX};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; a++)
      m[a] = min(m[a], xyz[3 * i + a]);
#pragma unroll
  for (int a = 0; a < 3; a++)
    for (int o = 32; o > 0; o >>= 1)
      m[a] = min(m[a], __shfl_xor(m[a], o));
  __shared__ int sm[3][4];
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int a = 0; a < 3; a++)
      sm[a][threadIdx.x >> 6] = m[a];
  __syncthreads();
  if (threadIdx.x < 3) {
    int v = sm[threadIdx.x][0];
    for (int t = 1; t < (int)(blockDim.x >> 6); t++)
      v = min(v, sm[threadIdx.x][t]);
    atomicMin(&mn[threadIdx.x], v);
  }
}

__global__ void shift_kernel(int32_t* __restrict__ xyz, int64_t n3, const int32_t* __restrict__ mn)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n3)
    xyz[i] -= mn[i % 3];  // pointCloud[i] -= box.min (TMC3.cpp:70-72)
}

__global__ void color_scatter_kernel(const PlaneRec* __restrict__ planes, const int32_t* __restrict__ list,
                                     const int32_t* __restrict__ rgb, int64_t n, uint16_t* __restrict__ colors)
{
  const PlaneRec r = planes[blockIdx.x];
  const uint16_t c0 = (uint16_t)rgb[3 * blockIdx.x], c1 = (uint16_t)rgb[3 * blockIdx.x + 1],
                 c2 = (uint16_t)rgb[3 * blockIdx.x + 2];
  for (int64_t t = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; t < r.list_n; t += (int64_t)gridDim.y * blockDim.x) {
    const int64_t id = list[r.list_off + t];
    if (id >= 0 && id < n) {
      colors[3 * id] = c0;
      colors[3 * id + 1] = c1;
      colors[3 * id + 2] = c2;
    }
  }
}

// PLY ingest on the device: one thread per vertex record of a binary
// little-endian body; position = (int32) trunc(value * scale) exactly as ply.cpp:436-465
// (float promoted to double, double product, C conversion toward zero), AoS int32 out.
// Records are byte-packed (27 B, 15 B, ...): fields are assembled from bytes.
__device__ inline double load_scalar_le(const unsigned char* p, int is_f64)
{
  if (is_f64) {
    unsigned long long b = 0;
#pragma unroll
    for (int t = 0; t < 8; t++)
      b |= (unsigned long long)p[t] << (8 * t);
    return __longlong_as_double((long long)b);
  }
  unsigned int b = 0;
#pragma unroll
  for (int t = 0; t < 4; t++)
    b |= (unsigned int)p[t] << (8 * t);
  return (double)__uint_as_float(b);
}

__global__ void ingest_kernel(const unsigned char* __restrict__ rec, int64_t n, int stride, int ox, int oy, int oz,
                              int is_f64, double scale, int32_t* __restrict__ xyz, int* __restrict__ bad)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const unsigned char* r = rec + i * stride;
  const double v[3] = {load_scalar_le(r + ox, is_f64) * scale, load_scalar_le(r + oy, is_f64) * scale,
                       load_scalar_le(r + oz, is_f64) * scale};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    // outside int32 the reference's conversion is undefined behaviour: report instead
    if (!(v[a] > -2147483649.0 && v[a] < 2147483648.0))
      *bad = 1;
    xyz[3 * i + a] = (int32_t)v[a];
  }
}

}  // namespace
}  // namespace bs

using namespace bs;

extern "C" int bs_shift_to_origin_dev(bs_ctx* ctx, int32_t* d_xyz, int64_t n, int32_t* min_out);

extern "C" int bs_ingest_dev(bs_ctx* ctx, const void* d_records, int64_t n, int32_t stride, int32_t off_x, int32_t off_y,
                             int32_t off_z, int32_t is_f64, double scale, int32_t shift_to_origin, int32_t* d_xyz,
                             int32_t* min_out)
{
  if (!ctx)
    return BS_ERR_INVALID;
  const int w = is_f64 ? 8 : 4;
  if (!d_records || !d_xyz || n <= 0 || stride < 3 * w || off_x < 0 || off_y < 0 || off_z < 0 || off_x + w > stride ||
      off_y + w > stride || off_z + w > stride)
    return fail(ctx, BS_ERR_INVALID, "bs_ingest_dev: null pointer or inconsistent record layout");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->misc.reserve(256));
  hipStream_t st = ctx->stream;
  int* d_bad = ctx->misc.as<int>() + 56;
  BS_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(int), st));
  ingest_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>((const unsigned char*)d_records, n, stride, off_x, off_y, off_z,
                                                      is_f64, scale, d_xyz, d_bad);
  int bad = 0;
  BS_HIP(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  if (bad)
    return fail(ctx, BS_ERR_RANGE, "bs_ingest_dev: value * scale does not fit int32 (undefined in the reference, ply.cpp:436-465)");
  if (shift_to_origin)
    return bs_shift_to_origin_dev(ctx, d_xyz, n, min_out);
  if (min_out)
    min_out[0] = min_out[1] = min_out[2] = 0;
  ctx->order_n = 0;
  return BS_OK;
}

extern "C" int bs_shift_to_origin_dev(bs_ctx* ctx, int32_t* d_xyz, int64_t n, int32_t* min_out)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_xyz || n <= 0)
    return fail(ctx, BS_ERR_INVALID, "null pointer or empty cloud");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->misc.reserve(256));
  hipStream_t st = ctx->stream;
  int32_t init[3] = {INT_MAX, INT_MAX, INT_MAX};
  int32_t* d_mn = ctx->misc.as<int32_t>() + 48;
  BS_HIP(ctx, hipMemcpyAsync(d_mn, init, sizeof init, hipMemcpyHostToDevice, st));
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 512);
  minmax_kernel<<<blocks, 256, 0, st>>>(d_xyz, n, d_mn);
  shift_kernel<<<(int)((3 * n + 255) / 256), 256, 0, st>>>(d_xyz, 3 * n, d_mn);
  int32_t h[3];
  BS_HIP(ctx, hipMemcpyAsync(h, d_mn, sizeof h, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  if (min_out)
    for (int a = 0; a < 3; a++)
      min_out[a] = h[a];
  ctx->order_n = 0;  // coordinates changed: a cached cell order no longer applies
  return BS_OK;
}

extern "C" int bs_plane_colors_dev(bs_ctx* ctx, const int32_t* plane_rgb, int32_t n_planes, int64_t n,
                                   uint16_t* d_colors)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_colors || n <= 0 || (n_planes > 0 && !plane_rgb))
    return fail(ctx, BS_ERR_INVALID, "null pointer");
  if (!ctx->rg_valid || ctx->rg_n != n)
    return fail(ctx, BS_ERR_INVALID, "no region-grow result for this cloud on the context");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  GrowStats hs;
  BS_HIP(ctx, hipMemcpy(&hs, ctx->rg_stats.p, sizeof hs, hipMemcpyDeviceToHost));
  if (hs.n_planes != n_planes)
    return fail(ctx, BS_ERR_INVALID, "n_planes does not match the last region grow");
  BS_HIP(ctx, hipMemsetAsync(d_colors, 0, sizeof(uint16_t) * 3 * n, st));  // my_function.cpp:262-264
  if (n_planes > 0) {
    BS_HIP(ctx, ctx->misc.reserve(256));
    BS_HIP(ctx, ctx->fb_list.reserve(sizeof(int32_t) * (3 * (size_t)n_planes + 16)));
    int32_t* d_rgb = ctx->fb_list.as<int32_t>();
    BS_HIP(ctx, hipMemcpyAsync(d_rgb, plane_rgb, sizeof(int32_t) * 3 * n_planes, hipMemcpyHostToDevice, st));
    color_scatter_kernel<<<dim3(n_planes, 16), 256, 0, st>>>(ctx->rg_planes.as<PlaneRec>(), ctx->rg_list.as<int32_t>(),
                                                          d_rgb, n, d_colors);
  }
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  return BS_OK;
}

namespace {
__global__ void center_div_selftest_kernel(const int32_t* __restrict__ c, const uint32_t* __restrict__ n,
                                           int32_t* __restrict__ out, int64_t count)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  const bs::CenterDiv d = bs::center_div_prepare(n[i]);
  out[i] = bs::center_div(c[i], d);
}
}  // namespace

extern "C" int bs_selftest_center_div(bs_ctx* ctx, const int32_t* c, const uint32_t* n, int32_t* out, int64_t count)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!c || !n || !out || count <= 0 || count > (1ll << 30))
    return fail(ctx, BS_ERR_INVALID, "null pointer or bad count");
  for (int64_t i = 0; i < count; i++)
    if (n[i] == 0 || n[i] >= 0x80000000u)
      return fail(ctx, BS_ERR_RANGE, "divisor outside 1 .. 2^31-1");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->misc.reserve((size_t)count * 12 + 256));
  hipStream_t st = ctx->stream;
  int32_t* d_c = ctx->misc.as<int32_t>() + 64;
  uint32_t* d_n = reinterpret_cast<uint32_t*>(d_c + count);
  int32_t* d_o = d_c + 2 * count;
  BS_HIP(ctx, hipMemcpyAsync(d_c, c, sizeof(int32_t) * count, hipMemcpyHostToDevice, st));
  BS_HIP(ctx, hipMemcpyAsync(d_n, n, sizeof(uint32_t) * count, hipMemcpyHostToDevice, st));
  center_div_selftest_kernel<<<(int)((count + 255) / 256), 256, 0, st>>>(d_c, d_n, d_o, count);
  BS_HIP(ctx, hipMemcpyAsync(out, d_o, sizeof(int32_t) * count, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  return BS_OK;
}
