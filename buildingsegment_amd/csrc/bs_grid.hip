// bs_grid.hip -- search-grid build for the kNN / normal kernels (gfx950).
//
// Replaces the two nanoflann kd-tree builds the reference triggers through
// Open3D (/root/reference/tmc3/my_function.h:63 and :71) by ONE hashed uniform
// grid: points are binned into cubic cells, radix-sorted by cell key (rocPRIM
// via hipCUB as a primitive), and each occupied cell gets a 16-byte entry
// {key, start, end} in an open-addressing table.  The cell-sorted copy of the
// cloud (int4 = x,y,z,global index) is what the query kernels stream, so all
// points of a cell are one contiguous, coalesced run in HBM.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "bs_common.h"

namespace bs {

namespace {

__global__ void bbox_kernel(const int32_t* __restrict__ xyz, int64_t n, int32_t* __restrict__ mnmx)
{
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      int v = xyz[3 * i + a];
      mn[a] = min(mn[a], v);
      mx[a] = max(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o));
    }
  }
  __shared__ int smn[3][4], smx[3][4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      smn[a][wv] = mn[a];
      smx[a][wv] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    int m0 = smn[a][0], m1 = smx[a][0];
    for (int t = 1; t < (int)(blockDim.x >> 6); t++) {
      m0 = min(m0, smn[a][t]);
      m1 = max(m1, smx[a][t]);
    }
    atomicMin(&mnmx[a], m0);
    atomicMax(&mnmx[3 + a], m1);
  }
}

// Sort key of a cell.  The hash table is keyed by pack_cell(); the ORDER of the cells in
// the sorted copy is free (only "points of one cell are contiguous" matters).  Morton
// (Z-curve) order keeps cells that are adjacent in ANY axis close in memory, so the
// gathers of everything that walks the cell-sorted order (kNN candidates, static masks,
// reverse lists, owner passes) find their neighbours' lines in the XCD's L2; the raster
// order (x fastest) separates y/z neighbours by a whole row / layer of the scene.
__host__ __device__ inline uint64_t spread21(uint64_t v)
{
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__host__ __device__ inline uint32_t compact21(uint64_t v)
{
  v &= 0x1249249249249249ull;
  v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
  v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
  v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
  v = (v | (v >> 16)) & 0x1F00000000FFFFull;
  v = (v | (v >> 32)) & 0x1FFFFFull;
  return (uint32_t)v;
}

__global__ void cellkey_kernel(const int32_t* __restrict__ xyz, int64_t n, int mnx, int mny, int mnz,
                               int cell, int morton, uint64_t* __restrict__ keys, int32_t* __restrict__ vals)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  uint32_t cx = (uint32_t)(xyz[3 * i] - mnx) / (uint32_t)cell;
  uint32_t cy = (uint32_t)(xyz[3 * i + 1] - mny) / (uint32_t)cell;
  uint32_t cz = (uint32_t)(xyz[3 * i + 2] - mnz) / (uint32_t)cell;
  keys[i] = morton ? (spread21(cx) | (spread21(cy) << 1) | (spread21(cz) << 2)) : pack_cell(cx, cy, cz);
  if (vals)
    vals[i] = (int32_t)i;
}

__global__ void count_heads_kernel(const uint64_t* __restrict__ keys, int64_t n, unsigned long long* cnt)
{
  unsigned int local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    local += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1)
    local += __shfl_xor(local, o);
  __shared__ unsigned int part[4];
  if ((threadIdx.x & 63) == 0)
    part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++)
      t += part[k];
    if (t)
      atomicAdd(cnt, (unsigned long long)t);
  }
}

__global__ void table_clear_kernel(CellEntry* t, uint32_t size)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size) {
    t[i].key = ~0ull;
    t[i].start = 0;
    t[i].end = 0;
  }
}

__global__ void table_insert_kernel(const uint64_t* __restrict__ ukeys, const int32_t* __restrict__ ucnt,
                                    const int32_t* __restrict__ ustart, int32_t ncell, int morton,
                                    CellEntry* table, uint32_t hmask)
{
  int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell)
    return;
  uint64_t k = ukeys[c];
  if (morton)
    k = pack_cell(compact21(k), compact21(k >> 1), compact21(k >> 2));
  uint32_t h = hash_cell(k) & hmask;
  for (;;) {
    unsigned long long prev =
        atomicCAS((unsigned long long*)&table[h].key, ~0ull, (unsigned long long)k);
    if (prev == ~0ull)
      break;
    h = (h + 1) & hmask;
  }
  table[h].start = ustart[c];
  table[h].end = ustart[c] + ucnt[c];
}

__global__ void gather_sorted_kernel(const int32_t* __restrict__ xyz, const int32_t* __restrict__ gidx,
                                     const int32_t* __restrict__ order, int64_t n, int4* __restrict__ spts)
{
  int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= n)
    return;
  int32_t i = order[s];
  int4 v;
  v.x = xyz[3 * (int64_t)i];
  v.y = xyz[3 * (int64_t)i + 1];
  v.z = xyz[3 * (int64_t)i + 2];
  v.w = gidx ? gidx[i] : i;
  spts[s] = v;
}

inline int grid_blocks(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

}  // namespace

// Occupied-cell count at a trial cell size (keys only).  (Counting by insertion into an open-addressing table instead
// of sorting was measured: the points arrive in the caller's order, 50 M random probes into a 1 GB table take 4.5 ms
// against 2.7 ms for the radix sort -- grid stage 19.7 instead of 11.0 ms.)
// Bits a Morton cell key occupies when the longest axis has ext / cell + 1 cells: the radix sorts run over these only
// (33 instead of 63 bits at 50 M points: 5 instead of 8 passes).
static int morton_key_bits(int64_t ext, int cell)
{
  int bits = 1;
  while (((int64_t)1 << bits) < ext / cell + 1)
    bits++;
  return std::min(63, 3 * bits);
}

static int count_cells(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, const int mn[3], int cell, int64_t ext,
                       int64_t* ncell)
{
  hipStream_t st = ctx->stream;
  uint64_t* kin = ctx->keys_in.as<uint64_t>();
  uint64_t* kout = ctx->keys_out.as<uint64_t>();
  const int end_bit = morton_key_bits(ext, cell);
  cellkey_kernel<<<grid_blocks(n, 256), 256, 0, st>>>(d_xyz, n, mn[0], mn[1], mn[2], cell, 1, kin, nullptr);
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, kin, kout, (int)n, 0, end_bit, st));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(ctx->cub_tmp.p, tb, kin, kout, (int)n, 0, end_bit, st));
  unsigned long long* cnt = ctx->misc.as<unsigned long long>() + 8;
  BS_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
  count_heads_kernel<<<std::min(grid_blocks(n, 256), 1024), 256, 0, st>>>(kout, n, cnt);
  unsigned long long h = 0;
  BS_HIP(ctx, hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  *ncell = (int64_t)h;
  return BS_OK;
}

int build_grid(bs_ctx* ctx, const int32_t* d_xyz, const int32_t* d_gidx, int64_t n, double radius,
               int k, int cell_hint, GridDev* out)
{
  hipStream_t st = ctx->stream;
  if (n <= 0 || n >= (int64_t)INT_MAX - 64)
    return fail(ctx, BS_ERR_INVALID, "point count out of range");
  BS_HIP(ctx, ctx->misc.reserve(256));
  BS_HIP(ctx, ctx->keys_in.reserve(sizeof(uint64_t) * n));
  BS_HIP(ctx, ctx->keys_out.reserve(sizeof(uint64_t) * n));
  BS_HIP(ctx, ctx->vals_in.reserve(sizeof(int32_t) * n));
  BS_HIP(ctx, ctx->vals_out.reserve(sizeof(int32_t) * n));

  // 1. bounding box
  int32_t init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
  int32_t* d_mnmx = ctx->misc.as<int32_t>();
  BS_HIP(ctx, hipMemcpyAsync(d_mnmx, init, sizeof init, hipMemcpyHostToDevice, st));
  bbox_kernel<<<std::min(grid_blocks(n, 256), 512), 256, 0, st>>>(d_xyz, n, d_mnmx);
  int32_t bb[6];
  BS_HIP(ctx, hipMemcpyAsync(bb, d_mnmx, sizeof bb, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  // exact-arithmetic domain: moment sums stay below 2^53, d^2 keys below 2^64
  const int32_t LIM = 1 << 23;
  for (int a = 0; a < 3; a++)
    if (bb[a] <= -LIM || bb[3 + a] >= LIM)
      return fail(ctx, BS_ERR_RANGE, "coordinates must satisfy |c| < 2^23 mm (8.4 km): shift the cloud to its bounding-box origin first "
                                     "(bs_shift_to_origin_dev, or the buildingSeg constructor as TMC3.cpp:210 does)");

  // 2. cell size: >= radius so one ring certifies the hybrid search, and a
  // density-driven size giving ~max(4, k/2) points per occupied cell
  int64_t ext = 1;
  for (int a = 0; a < 3; a++)
    ext = std::max<int64_t>(ext, (int64_t)bb[3 + a] - bb[a] + 1);
  int cell = cell_hint > 0 ? cell_hint : (int)std::max(1.0, std::ceil(radius));
  if (cell_hint <= 0) {
    const double target = std::max(4.0, 0.5 * k);
    const int cmin = cell;
    for (int it = 0; it < 4; it++) {
      int64_t nc = 0;
      int rc = count_cells(ctx, d_xyz, n, bb, cell, ext, &nc);
      if (rc != BS_OK)
        return rc;
      double occ = (double)n / (double)std::max<int64_t>(nc, 1);
      if (occ >= 0.7 * target && occ <= 1.6 * target)
        break;
      double f = std::sqrt(target / occ);
      f = std::min(4.0, std::max(0.25, f));
      int ncell = (int)std::floor(cell * f + 0.5);
      ncell = std::max(ncell, cmin);
      if (ncell == cell || (int64_t)ncell > ext)
        break;
      cell = ncell;
    }
  }
  while (ext / cell + 1 >= (1 << 21))
    cell *= 2;

  // 3. keys + sort
  uint64_t* kin = ctx->keys_in.as<uint64_t>();
  uint64_t* kout = ctx->keys_out.as<uint64_t>();
  int32_t* vin = ctx->vals_in.as<int32_t>();
  int32_t* vout = ctx->vals_out.as<int32_t>();
  static const int morton = []() {
    const char* e = getenv("BS_GRID_ORDER");  // developer A/B switch: "raster" restores the x-fastest cell order
    return (e && e[0] == 'r') ? 0 : 1;
  }();
  cellkey_kernel<<<grid_blocks(n, 256), 256, 0, st>>>(d_xyz, n, bb[0], bb[1], bb[2], cell, morton, kin, vin);
  size_t tb = 0;
  const int end_bit = morton ? morton_key_bits(ext, cell) : 63;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, end_bit, st));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tb, kin, kout, vin, vout, (int)n, 0, end_bit, st));

  // 4. unique cells (run-length encode), starts (exclusive scan)
  BS_HIP(ctx, ctx->uniq_keys.reserve(sizeof(uint64_t) * n));
  BS_HIP(ctx, ctx->uniq_cnt.reserve(sizeof(int32_t) * 2 * n));
  uint64_t* ukeys = ctx->uniq_keys.as<uint64_t>();
  int32_t* ucnt = ctx->uniq_cnt.as<int32_t>();
  int32_t* ustart = ucnt + n;
  int32_t* d_nruns = ctx->misc.as<int32_t>() + 32;
  tb = 0;
  BS_HIP(ctx, hipcub::DeviceRunLengthEncode::Encode(nullptr, tb, kout, ukeys, ucnt, d_nruns, (int)n, st));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRunLengthEncode::Encode(ctx->cub_tmp.p, tb, kout, ukeys, ucnt, d_nruns, (int)n, st));
  int32_t nruns = 0;
  BS_HIP(ctx, hipMemcpyAsync(&nruns, d_nruns, sizeof nruns, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  if (nruns <= 0)
    return fail(ctx, BS_ERR_INTERNAL, "grid: no occupied cells");
  tb = 0;
  BS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ucnt, ustart, nruns, st));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->cub_tmp.p, tb, ucnt, ustart, nruns, st));

  // 5. hash table
  uint32_t hs = 64;
  while (hs < (uint32_t)nruns * 2u)
    hs <<= 1;
  BS_HIP(ctx, ctx->table.reserve(sizeof(CellEntry) * (size_t)hs));
  CellEntry* table = ctx->table.as<CellEntry>();
  table_clear_kernel<<<(hs + 255) / 256, 256, 0, st>>>(table, hs);
  table_insert_kernel<<<(nruns + 255) / 256, 256, 0, st>>>(ukeys, ucnt, ustart, nruns, morton, table, hs - 1);

  // 6. cell-sorted point copy
  BS_HIP(ctx, ctx->spts.reserve(sizeof(int4) * n));
  gather_sorted_kernel<<<grid_blocks(n, 256), 256, 0, st>>>(d_xyz, d_gidx, vout, n, ctx->spts.as<int4>());
  BS_HIP(ctx, hipGetLastError());

  out->mn[0] = bb[0];
  out->mn[1] = bb[1];
  out->mn[2] = bb[2];
  for (int a = 0; a < 3; a++)
    out->dim[a] = (int32_t)(((int64_t)bb[3 + a] - bb[a]) / cell + 1);
  out->cell = cell;
  out->hmask = hs - 1;
  out->table = table;
  out->spts = ctx->spts.as<int4>();
  out->slocal = vout;
  out->n = n;
  ctx->order_n = n;  // vals_out holds the cell-sorted order of this cloud
  ctx->order_xyz = d_xyz;
  return BS_OK;
}

// bounding box of a device-resident cloud (host result; synchronises): {min x,y,z, max x,y,z}
int bbox_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, int32_t bb[6])
{
  hipStream_t st = ctx->stream;
  BS_HIP(ctx, ctx->misc.reserve(256));
  int32_t init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
  int32_t* d_mnmx = ctx->misc.as<int32_t>();
  BS_HIP(ctx, hipMemcpyAsync(d_mnmx, init, sizeof init, hipMemcpyHostToDevice, st));
  if (n > 0)
    bbox_kernel<<<std::min(grid_blocks(n, 256), 512), 256, 0, st>>>(d_xyz, n, d_mnmx);
  BS_HIP(ctx, hipMemcpyAsync(bb, d_mnmx, 6 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  return BS_OK;
}

// Spatial (Morton) order of a cloud WITHOUT a search grid: what the region grower needs when it is handed
// foreign buffers (bs_region_grow[_dev], the component-sharded stage 3) -- its records, masks, reverse lists and
// owner passes address points by their position in a spatially coherent order (bs_grow_spec.hip, "Index
// spaces"); with the identity order every neighbour is a random HBM access.  One bounding-box pass, one key
// pass, one radix sort over exactly the key bits in use.  Any int32 coordinates are admissible (no exact-moment
// domain here: only an order is produced).  Result: ctx->vals_out = original index of every position.
int build_spatial_order(bs_ctx* ctx, const int32_t* d_xyz, int64_t n)
{
  hipStream_t st = ctx->stream;
  if (n <= 0 || n >= (int64_t)INT_MAX - 64)
    return fail(ctx, BS_ERR_INVALID, "point count out of range");
  BS_HIP(ctx, ctx->misc.reserve(256));
  BS_HIP(ctx, ctx->keys_in.reserve(sizeof(uint64_t) * n));
  BS_HIP(ctx, ctx->keys_out.reserve(sizeof(uint64_t) * n));
  BS_HIP(ctx, ctx->vals_in.reserve(sizeof(int32_t) * n));
  BS_HIP(ctx, ctx->vals_out.reserve(sizeof(int32_t) * n));
  int32_t init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
  int32_t* d_mnmx = ctx->misc.as<int32_t>();
  BS_HIP(ctx, hipMemcpyAsync(d_mnmx, init, sizeof init, hipMemcpyHostToDevice, st));
  bbox_kernel<<<std::min(grid_blocks(n, 256), 512), 256, 0, st>>>(d_xyz, n, d_mnmx);
  int32_t bb[6];
  BS_HIP(ctx, hipMemcpyAsync(bb, d_mnmx, sizeof bb, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  int64_t ext = 1;
  for (int a = 0; a < 3; a++)
    ext = std::max<int64_t>(ext, (int64_t)bb[3 + a] - bb[a] + 1);
  // cells of >= 128 mm (a few points each at the densities of SURVEY 8(d)): fewer key bits = fewer radix passes
  int64_t cell = 128;
  while (ext / cell + 1 >= (1 << 21))
    cell *= 2;
  int bits = 1;
  while (((int64_t)1 << bits) < ext / cell + 1)
    bits++;
  uint64_t* kin = ctx->keys_in.as<uint64_t>();
  uint64_t* kout = ctx->keys_out.as<uint64_t>();
  int32_t* vin = ctx->vals_in.as<int32_t>();
  int32_t* vout = ctx->vals_out.as<int32_t>();
  cellkey_kernel<<<grid_blocks(n, 256), 256, 0, st>>>(d_xyz, n, bb[0], bb[1], bb[2], (int)cell, 1, kin, vin);
  const int end_bit = std::min(63, 3 * bits);
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, end_bit, st));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tb, kin, kout, vin, vout, (int)n, 0, end_bit, st));
  BS_HIP(ctx, hipGetLastError());
  ctx->order_n = n;
  ctx->order_xyz = d_xyz;
  return BS_OK;
}

}  // namespace bs
