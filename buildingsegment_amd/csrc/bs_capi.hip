// bs_capi.hip -- extern "C" boundary of libbuildingsegment_hip.so
// (declarations and reference citations: include/bs_api.h).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "bs_common.h"

namespace bs {

int fail(bs_ctx* ctx, int status, const char* what, hipError_t e)
{
  if (ctx) {
    ctx->err = what ? what : "";
    if (e != hipSuccess) {
      ctx->err += ": ";
      ctx->err += hipGetErrorString(e);
    }
  }
  return status;
}

static int check_params(bs_ctx* ctx, const bs_params* p, int64_t n)
{
  if (!p)
    return fail(ctx, BS_ERR_INVALID, "params is NULL");
  if (p->k < 2 || p->k > 32)
    return fail(ctx, BS_ERR_INVALID, "k must be in [2, 32]");
  if (p->max_nn < 3 || p->max_nn > 64)
    return fail(ctx, BS_ERR_INVALID, "max_nn must be in [3, 64]");
  if (!(p->radius > 0.0) || p->radius > 1.0e6)
    return fail(ctx, BS_ERR_INVALID, "radius must be in (0, 1e6]");
  if (p->th_thickness < 0 || p->th_point_count < 0)
    return fail(ctx, BS_ERR_INVALID, "thresholds must be >= 0");
  if (n < p->k)
    return fail(ctx, BS_ERR_INVALID, "n < k (the reference is undefined there)");
  if (n >= (int64_t)INT32_MAX - 64)
    return fail(ctx, BS_ERR_INVALID, "n must fit in int32");
  if (p->rg_mode < 0 || p->rg_mode > 2)
    return fail(ctx, BS_ERR_INVALID, "rg_mode must be 0, 1 or 2");
  return BS_OK;
}

struct EventTimer {
  bs_ctx* ctx;
  explicit EventTimer(bs_ctx* c) : ctx(c) {}
  void mark(int i) { (void)hipEventRecord(ctx->ev[i], ctx->stream); }
  double ms(int i, int j)
  {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[j]) != hipSuccess)
      return 0.0;
    return (double)t;
  }
};

// foreign neighbour arrays are dereferenced on the device: one streaming pass proves them in range
__global__ void neigh_range_kernel(const int32_t* __restrict__ neigh, int64_t total, int32_t n, int* bad)
{
  bool b = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = neigh[i];
    b = b || v < 0 || v >= n;
  }
  if (__syncthreads_or(b) && threadIdx.x == 0)
    *bad = 1;
}

static int region_grow_dev_impl(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals, const int32_t* d_neigh,
                                int64_t n, const bs_params* p, int32_t* d_plane_idx, bool trusted_neigh)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_xyz || !d_normals || !d_neigh || !d_plane_idx)
    return fail(ctx, BS_ERR_INVALID, "null device pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  if (!trusted_neigh) {
    BS_HIP(ctx, ctx->misc.reserve(256));
    int* d_bad = ctx->misc.as<int>() + 48;
    BS_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
    const int64_t total = n * (int64_t)p->k;
    neigh_range_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 65536), 256, 0, ctx->stream>>>(d_neigh, total, (int32_t)n, d_bad);
    int bad = 0;
    BS_HIP(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    BS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad)
      return fail(ctx, BS_ERR_INVALID, "neighbour index out of range");
  }
  EventTimer T(ctx);
  T.mark(3);
  if (p->rg_mode == 1)
    rc = launch_region_grow_seq(ctx, d_xyz, d_normals, d_neigh, n, *p, d_plane_idx);
  else
    rc = launch_region_grow_spec(ctx, d_xyz, d_normals, d_neigh, n, *p, d_plane_idx);
  if (rc != BS_OK)
    return rc;
  T.mark(4);
  BS_HIP(ctx, hipEventSynchronize(ctx->ev[4]));
  ctx->tm.grow_ms = T.ms(3, 4);
  return BS_OK;
}

}  // namespace bs

using namespace bs;

extern "C" {

int bs_api_version(void) { return BS_API_VERSION; }

int64_t bs_sizeof_timings(void) { return (int64_t)sizeof(bs_timings); }

const char* bs_strerror(int status)
{
  switch (status) {
  case BS_OK: return "ok";
  case BS_ERR_INVALID: return "invalid argument";
  case BS_ERR_RANGE: return "coordinate outside the exact domain (|c| < 2^23 mm): shift the cloud to its bounding-box origin first (bs_shift_to_origin_dev / the buildingSeg constructor)";
  case BS_ERR_NOMEM: return "out of memory";
  case BS_ERR_HIP: return "HIP runtime error";
  case BS_ERR_NO_DEVICE: return "no usable HIP device";
  case BS_ERR_INTERNAL: return "internal invariant violated";
  case BS_ERR_UNCERTIFIED: return "halo too thin to certify every k-list";
  default: return "unknown status";
  }
}

void bs_params_default(bs_params* p)
{
  if (!p)
    return;
  p->k = 15;              // TMC3.cpp:215-216
  p->max_nn = 50;         // my_function.h:63
  p->radius = 100.0;      // my_function.h:63
  p->th_thickness = 300;  // my_function.h:117
  p->th_point_count = 400;  // my_function.h:118
  p->cos_th = 0.88;       // my_function.cpp:230
  p->cell_size = 0;
  p->rg_mode = 0;
}

int bs_create(int device, bs_ctx** out)
{
  if (!out)
    return BS_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return BS_ERR_NO_DEVICE;
  if (device < 0 || device >= count)
    return BS_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess)
    return BS_ERR_NO_DEVICE;
  bs_ctx* c = new (std::nothrow) bs_ctx();
  if (!c)
    return BS_ERR_NOMEM;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return BS_ERR_HIP;
  }
  c->stream = c->own_stream;
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      bs_destroy(c);
      return BS_ERR_HIP;
    }
  *out = c;
  return BS_OK;
}

void bs_destroy(bs_ctx* c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  bs::DevBuf* bufs[] = {&c->keys_in, &c->keys_out, &c->vals_in, &c->vals_out, &c->cub_tmp, &c->uniq_keys,
                        &c->uniq_cnt, &c->misc, &c->table, &c->spts, &c->slocal, &c->fb_list, &c->d_xyz_h,
                        &c->d_neigh_h, &c->d_normals_h, &c->d_plane_h, &c->seg_neigh, &c->seg_normals,
                        &c->rg_list, &c->rg_stack, &c->rg_planes, &c->rg_stats, &c->rg_aux, &c->rg_pstore, &c->rg_rec, &c->rg_radj, &c->rg_roff, &c->rg_geo, &c->rg_gs, &c->rg_disp, &c->seg_npos,
                        &c->rs_keys_in, &c->rs_keys_out, &c->rs_vals_in, &c->rs_vals_out, &c->rs_cnt, &c->rs_img, &c->rs_tmp};
  for (auto* b : bufs)
    b->release();
  for (auto& b : c->sh)
    b.release();
  c->rg_hout.release();
  for (auto& e : c->ev)
    if (e)
      (void)hipEventDestroy(e);
  for (auto& e : c->sev)
    if (e)
      (void)hipEventDestroy(e);
  if (c->side)
    (void)hipStreamDestroy(c->side);
  if (c->own_stream)
    (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* bs_last_error(const bs_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bs_set_stream(bs_ctx* ctx, void* hip_stream)
{
  if (!ctx)
    return BS_ERR_INVALID;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return BS_OK;
}

int bs_get_timings(const bs_ctx* ctx, bs_timings* out)
{
  if (!ctx || !out)
    return BS_ERR_INVALID;
  *out = ctx->tm;
  return BS_OK;
}

// ---------------------------------------------------------------------------
// device-buffer entry points
// ---------------------------------------------------------------------------

static int knn_normals_dev_impl(bs_ctx* ctx, const int32_t* d_xyz, const int32_t* d_gidx, int64_t n, int64_t q_begin,
                                int64_t q_end, const bs_params* p, int32_t* d_neigh, double* d_normals,
                                double cert_radius, int64_t* n_uncertified, int32_t* d_npos)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_xyz || !d_neigh)
    return fail(ctx, BS_ERR_INVALID, "null device pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  if (q_begin < 0 || q_end > n || q_begin > q_end)
    return fail(ctx, BS_ERR_INVALID, "bad query range");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  EventTimer T(ctx);
  T.mark(0);
  GridDev g;
  rc = build_grid(ctx, d_xyz, d_gidx, n, p->radius, p->k, p->cell_size, &g);
  if (rc != BS_OK)
    return rc;
  T.mark(1);
  ctx->npos_neigh = nullptr;
  ctx->npos_normals = nullptr;
  rc = launch_knn_normals(ctx, g, q_begin, q_end, *p, d_neigh, d_normals, cert_radius, n_uncertified, d_npos);
  if (rc != BS_OK)
    return rc;
  T.mark(2);
  BS_HIP(ctx, hipEventSynchronize(ctx->ev[2]));
  ctx->tm.grid_ms = T.ms(0, 1);
  ctx->tm.knn_ms = T.ms(1, 2);
  ctx->tm.total_ms = T.ms(0, 2);
  if (d_npos) {  // valid for exactly this neighbour buffer / k (checked by the grower)
    ctx->npos_neigh = d_neigh;
    ctx->npos_normals = d_normals;
    ctx->npos_k = p->k;
  }
  return BS_OK;
}

int bs_knn_normals_dev(bs_ctx* ctx, const int32_t* d_xyz, const int32_t* d_gidx, int64_t n, int64_t q_begin,
                       int64_t q_end, const bs_params* p, int32_t* d_neigh, double* d_normals,
                       double cert_radius, int64_t* n_uncertified)
{
  return knn_normals_dev_impl(ctx, d_xyz, d_gidx, n, q_begin, q_end, p, d_neigh, d_normals, cert_radius, n_uncertified,
                              nullptr);
}

int bs_region_grow_dev(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals, const int32_t* d_neigh,
                       int64_t n, const bs_params* p, int32_t* d_plane_idx)
{
  if (ctx) {  // the position-ordered copies of the fused pipeline are keyed by pointer, and these buffers' CONTENTS may
    ctx->npos_neigh = nullptr;  // have changed since: only bs_segment[_dev] itself hands them on to the grower
    ctx->npos_normals = nullptr;
  }
  return region_grow_dev_impl(ctx, d_xyz, d_normals, d_neigh, n, p, d_plane_idx, false);
}

int bs_segment_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, const bs_params* p, int32_t* d_neigh,
                   double* d_normals, int32_t* d_plane_idx)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_xyz || !d_plane_idx)
    return fail(ctx, BS_ERR_INVALID, "null device pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  if (!d_neigh) {
    BS_HIP(ctx, ctx->seg_neigh.reserve(sizeof(int32_t) * n * p->k));
    d_neigh = ctx->seg_neigh.as<int32_t>();
  }
  if (!d_normals) {
    BS_HIP(ctx, ctx->seg_normals.reserve(sizeof(double) * n * 3));
    d_normals = ctx->seg_normals.as<double>();
  }
  EventTimer T(ctx);
  T.mark(5);
  // the fused pipeline also keeps every neighbour's cell-sorted position for the grower (speculative mode)
  int32_t* d_npos = nullptr;
  if (p->rg_mode != 1) {
    BS_HIP(ctx, ctx->seg_npos.reserve(sizeof(int32_t) * ((size_t)n * p->k + 2) + sizeof(double) * 3 * (size_t)n));  // + normals by position
    d_npos = ctx->seg_npos.as<int32_t>();
  }
  rc = knn_normals_dev_impl(ctx, d_xyz, nullptr, n, 0, n, p, d_neigh, d_normals, 0.0, nullptr, d_npos);
  if (rc != BS_OK)
    return rc;
  rc = region_grow_dev_impl(ctx, d_xyz, d_normals, d_neigh, n, p, d_plane_idx, true);  // our own k-lists: in range
  if (rc != BS_OK)
    return rc;
  ctx->tm.total_ms = T.ms(5, 4);
  return BS_OK;
}

int bs_selftest_forge_next(bs_ctx* ctx, int mode)
{
  if (!ctx || mode < 0 || mode > 2)
    return BS_ERR_INVALID;
  ctx->forge_mode = mode;
  return BS_OK;
}

int bs_set_audit(bs_ctx* ctx, int on)
{
  if (!ctx)
    return BS_ERR_INVALID;
  ctx->audit = on ? 1 : 0;
  return BS_OK;
}

int bs_planes_fetch(bs_ctx* ctx, bs_planes* out)
{
  if (!ctx || !out)
    return BS_ERR_INVALID;
  memset(out, 0, sizeof *out);
  if (!ctx->rg_valid)
    return fail(ctx, BS_ERR_INVALID, "no region-grow result on this context");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  GrowStats hs;
  BS_HIP(ctx, hipMemcpy(&hs, ctx->rg_stats.p, sizeof hs, hipMemcpyDeviceToHost));
  const int np = hs.n_planes;
  out->n_planes = np;
  out->id = (int32_t*)malloc(sizeof(int32_t) * std::max(np, 1));
  out->normal = (double*)malloc(sizeof(double) * 3 * std::max(np, 1));
  out->center = (int32_t*)malloc(sizeof(int32_t) * 3 * std::max(np, 1));
  out->offset = (int64_t*)malloc(sizeof(int64_t) * (np + 1));
  out->point_idx = (int32_t*)malloc(sizeof(int32_t) * std::max<int64_t>(hs.list_used, 1));
  PlaneRec* recs = (PlaneRec*)malloc(sizeof(PlaneRec) * std::max(np, 1));
  if (!out->id || !out->normal || !out->center || !out->offset || !out->point_idx || !recs) {
    free(recs);
    bs_planes_free(out);
    return fail(ctx, BS_ERR_NOMEM, "host allocation failed");
  }
  hipError_t he = hipSuccess;
  if (np > 0)
    he = hipMemcpy(recs, ctx->rg_planes.p, sizeof(PlaneRec) * np, hipMemcpyDeviceToHost);
  if (he == hipSuccess && hs.list_used > 0)
    he = hipMemcpy(out->point_idx, ctx->rg_list.p, sizeof(int32_t) * hs.list_used, hipMemcpyDeviceToHost);
  if (he != hipSuccess) {
    free(recs);
    bs_planes_free(out);
    return fail(ctx, BS_ERR_HIP, "bs_planes_fetch: copy of the plane records failed", he);
  }
  for (int i = 0; i < np; i++) {
    out->id[i] = recs[i].id;
    for (int a = 0; a < 3; a++) {
      out->normal[3 * i + a] = recs[i].normal[a];
      out->center[3 * i + a] = recs[i].center[a];
    }
    out->offset[i] = recs[i].list_off;
  }
  out->offset[np] = hs.list_used;
  free(recs);
  return BS_OK;
}

void bs_planes_free(bs_planes* p)
{
  if (!p)
    return;
  free(p->id);
  free(p->normal);
  free(p->center);
  free(p->offset);
  free(p->point_idx);
  memset(p, 0, sizeof *p);
}

int bs_plane_colors(const bs_planes* planes, const int32_t* plane_rgb, int64_t n, uint16_t* colors)
{
  if (!planes || !colors || n < 0 || (planes->n_planes > 0 && !plane_rgb))
    return BS_ERR_INVALID;
  memset(colors, 0, sizeof(uint16_t) * 3 * (size_t)n);  // my_function.cpp:262-264
  for (int p = 0; p < planes->n_planes; p++) {          // :268-274
    for (int64_t t = planes->offset[p]; t < planes->offset[p + 1]; t++) {
      const int32_t id = planes->point_idx[t];
      if (id < 0 || id >= n)
        return BS_ERR_INVALID;
      for (int a = 0; a < 3; a++)
        colors[3 * (int64_t)id + a] = (uint16_t)plane_rgb[3 * p + a];
    }
  }
  return BS_OK;
}

// ---------------------------------------------------------------------------
// host-buffer entry points (stage through context-owned device buffers)
// ---------------------------------------------------------------------------

int bs_knn_normals(bs_ctx* ctx, const int32_t* xyz, int64_t n, const bs_params* p, int32_t* neigh,
                   double* normals)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!xyz || !neigh || !normals)
    return fail(ctx, BS_ERR_INVALID, "null host pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->d_xyz_h.reserve(sizeof(int32_t) * 3 * n));
  BS_HIP(ctx, ctx->d_neigh_h.reserve(sizeof(int32_t) * n * p->k));
  BS_HIP(ctx, ctx->d_normals_h.reserve(sizeof(double) * 3 * n));
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_xyz_h.p, xyz, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
  rc = bs_knn_normals_dev(ctx, ctx->d_xyz_h.as<int32_t>(), nullptr, n, 0, n, p, ctx->d_neigh_h.as<int32_t>(),
                          ctx->d_normals_h.as<double>(), 0.0, nullptr);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipMemcpyAsync(neigh, ctx->d_neigh_h.p, sizeof(int32_t) * n * p->k, hipMemcpyDeviceToHost, ctx->stream));
  BS_HIP(ctx, hipMemcpyAsync(normals, ctx->d_normals_h.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
  BS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BS_OK;
}

int bs_knn_normals_halo(bs_ctx* ctx, const int32_t* xyz, const int32_t* gidx, int64_t n, int64_t n_query,
                        const bs_params* p, int32_t* neigh, double* normals, double cert_radius,
                        int64_t* n_uncertified)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!xyz || !gidx || !neigh || !normals)
    return fail(ctx, BS_ERR_INVALID, "null host pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  if (n_query < 0 || n_query > n)
    return fail(ctx, BS_ERR_INVALID, "bad n_query");
  for (int64_t i = 0; i < n; i++)
    if (gidx[i] < 0)
      return fail(ctx, BS_ERR_INVALID, "negative global index");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->d_xyz_h.reserve(sizeof(int32_t) * 3 * n));
  BS_HIP(ctx, ctx->d_plane_h.reserve(sizeof(int32_t) * n));  // staging for gidx
  BS_HIP(ctx, ctx->d_neigh_h.reserve(sizeof(int32_t) * std::max<int64_t>(n_query, 1) * p->k));
  BS_HIP(ctx, ctx->d_normals_h.reserve(sizeof(double) * 3 * std::max<int64_t>(n_query, 1)));
  hipStream_t st = ctx->stream;
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_xyz_h.p, xyz, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice, st));
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_plane_h.p, gidx, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
  rc = bs_knn_normals_dev(ctx, ctx->d_xyz_h.as<int32_t>(), ctx->d_plane_h.as<int32_t>(), n, 0, n_query, p,
                          ctx->d_neigh_h.as<int32_t>(), ctx->d_normals_h.as<double>(), cert_radius, n_uncertified);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipMemcpyAsync(neigh, ctx->d_neigh_h.p, sizeof(int32_t) * n_query * p->k, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipMemcpyAsync(normals, ctx->d_normals_h.p, sizeof(double) * 3 * n_query, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  return BS_OK;
}

int bs_region_grow(bs_ctx* ctx, const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                   const bs_params* p, int32_t* plane_idx, bs_planes* planes)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (planes)
    memset(planes, 0, sizeof *planes);
  if (!xyz || !normals || !neigh || !plane_idx)
    return fail(ctx, BS_ERR_INVALID, "null host pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  // neighbour indices are dereferenced on the device: validate them here
  for (int64_t i = 0; i < n * (int64_t)p->k; i++)
    if (neigh[i] < 0 || neigh[i] >= n)
      return fail(ctx, BS_ERR_INVALID, "neighbour index out of range");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->d_xyz_h.reserve(sizeof(int32_t) * 3 * n));
  BS_HIP(ctx, ctx->d_neigh_h.reserve(sizeof(int32_t) * n * p->k));
  BS_HIP(ctx, ctx->d_normals_h.reserve(sizeof(double) * 3 * n));
  BS_HIP(ctx, ctx->d_plane_h.reserve(sizeof(int32_t) * n));
  hipStream_t st = ctx->stream;
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_xyz_h.p, xyz, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice, st));
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_neigh_h.p, neigh, sizeof(int32_t) * n * p->k, hipMemcpyHostToDevice, st));
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_normals_h.p, normals, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  ctx->npos_neigh = nullptr;  // the staging buffers now hold the caller's data, not what a previous bs_segment searched
  ctx->npos_normals = nullptr;
  rc = region_grow_dev_impl(ctx, ctx->d_xyz_h.as<int32_t>(), ctx->d_normals_h.as<double>(),
                            ctx->d_neigh_h.as<int32_t>(), n, p, ctx->d_plane_h.as<int32_t>(), true);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipMemcpyAsync(plane_idx, ctx->d_plane_h.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  if (planes)
    return bs_planes_fetch(ctx, planes);
  return BS_OK;
}

int bs_segment(bs_ctx* ctx, const int32_t* xyz, int64_t n, const bs_params* p, int32_t* neigh, double* normals,
               int32_t* plane_idx, bs_planes* planes)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (planes)
    memset(planes, 0, sizeof *planes);
  if (!xyz || !plane_idx)
    return fail(ctx, BS_ERR_INVALID, "null host pointer");
  int rc = check_params(ctx, p, n);
  if (rc != BS_OK)
    return rc;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->d_xyz_h.reserve(sizeof(int32_t) * 3 * n));
  BS_HIP(ctx, ctx->d_neigh_h.reserve(sizeof(int32_t) * n * p->k));
  BS_HIP(ctx, ctx->d_normals_h.reserve(sizeof(double) * 3 * n));
  BS_HIP(ctx, ctx->d_plane_h.reserve(sizeof(int32_t) * n));
  hipStream_t st = ctx->stream;
  BS_HIP(ctx, hipMemcpyAsync(ctx->d_xyz_h.p, xyz, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice, st));
  rc = bs_segment_dev(ctx, ctx->d_xyz_h.as<int32_t>(), n, p, ctx->d_neigh_h.as<int32_t>(),
                      ctx->d_normals_h.as<double>(), ctx->d_plane_h.as<int32_t>());
  if (rc != BS_OK)
    return rc;
  if (neigh)
    BS_HIP(ctx, hipMemcpyAsync(neigh, ctx->d_neigh_h.p, sizeof(int32_t) * n * p->k, hipMemcpyDeviceToHost, st));
  if (normals)
    BS_HIP(ctx, hipMemcpyAsync(normals, ctx->d_normals_h.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipMemcpyAsync(plane_idx, ctx->d_plane_h.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  if (planes)
    return bs_planes_fetch(ctx, planes);
  return BS_OK;
}

}  // extern "C"
