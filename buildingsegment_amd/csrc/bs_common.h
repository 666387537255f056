// bs_common.h -- shared declarations of the HIP product library
// (libbuildingsegment_hip.so).  gfx950 only; no CUDA, no dual paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/bs_api.h"

namespace bs {

// ---- small RAII-free device buffer that only grows -----------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap)
      return hipSuccess;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess)
      cap = want;
    return e;
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const
  {
    return (T*)p;
  }
};

// page-locked host staging (device-to-host results the host loop reads every round: a pageable
// destination goes through the runtime's bounce buffer in chunks, with the GPU idle in between)
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap)
      return hipSuccess;
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess)
      cap = want;
    return e;
  }
  void release()
  {
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const
  {
    return (T*)p;
  }
};

// ---- search grid (hashed uniform cells over the cell-sorted cloud) --------
struct CellEntry {  // 16 B: one dwordx4 per probe
  uint64_t key;     // packed cell coords, ~0 = empty slot
  int32_t start;    // first sorted position of the cell
  int32_t end;      // one past the last
};

struct GridDev {
  int32_t mn[3];       // bbox min (mm)
  int32_t dim[3];      // cells per axis (< 2^21)
  int32_t cell;        // cell edge (mm)
  uint32_t hmask;      // table size - 1
  const CellEntry* table;
  const int4* spts;    // cell-sorted points: x, y, z, global index
  const int32_t* slocal;  // cell-sorted local (input-order) index
  int64_t n;
};

// The fused pipeline's position-ordered scratch (bs_ctx::seg_npos): n * K neighbour positions, then the n normals
// in POSITION order (the grower builds its records by position and would otherwise gather 24 B per point by
// original index).
__host__ __device__ inline double* pnorm_of(int32_t* npos, int64_t n, int K)
{
  return reinterpret_cast<double*>(npos + (((int64_t)n * K + 1) & ~(int64_t)1));
}

__host__ __device__ inline uint64_t pack_cell(uint32_t cx, uint32_t cy, uint32_t cz)
{
  return (uint64_t)cx | ((uint64_t)cy << 21) | ((uint64_t)cz << 42);
}

__host__ __device__ inline uint32_t hash_cell(uint64_t k)
{
  k *= 0x9E3779B97F4A7C15ull;
  return (uint32_t)(k >> 32) ^ (uint32_t)k;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  For
// kernels that walk the cell-sorted order, map hardware block b to logical
// block (b % 8) * (nb / 8) + b / 8 (grid padded to a multiple of 8) so that one
// XCD sees one contiguous slice of space and spatial neighbours share its L2
// (speed only; any mapping is correct).
__device__ inline int64_t xcd_logical_block()
{
  const int64_t b = blockIdx.x, per = gridDim.x >> 3;
  return (b & 7) * per + (b >> 3);
}
inline int xcd_grid(int64_t nblocks) { return (int)((nblocks + 7) & ~(int64_t)7); }

// ---- plane record produced by region growing ------------------------------
struct PlaneRec {
  double normal[3];
  int64_t list_off;   // offset of pointIdx in the list pool
  int64_t list_n;     // pointIdx.size()
  int32_t center[3];
  int32_t id;         // 1-based plane id
  int32_t seed;
  int32_t pad;
};

struct GrowStats {
  int32_t n_planes;
  int32_t error;        // != 0: pool overflow / watchdog
  int64_t list_used;
  int64_t seed_attempts;
  int64_t largest;
  int64_t steps;
};

}  // namespace bs

// ---- context ---------------------------------------------------------------
struct bs_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  bs_timings tm{};
  hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

  // grid scratch
  bs::DevBuf keys_in, keys_out, vals_in, vals_out, cub_tmp, uniq_keys, uniq_cnt, misc;
  bs::DevBuf table, spts, slocal;
  // kNN scratch
  bs::DevBuf fb_list, d_xyz_h, d_neigh_h, d_normals_h, d_plane_h;
  // pipeline scratch (bs_segment_dev with NULL outputs)
  bs::DevBuf seg_neigh, seg_normals;
  // region-grow state
  bs::DevBuf rg_list, rg_stack, rg_planes, rg_stats, rg_aux, rg_pstore, rg_rec, rg_radj, rg_roff, rg_geo, rg_gs, rg_disp;
  bs::HostBuf rg_hout;  // PlaneOut[wave_cap + MAX_PENDING] + a few scalars, page-locked
  int64_t rg_n = 0;
  bool rg_valid = false;
  // final owner structure of the last speculative grow (inside rg_aux): owner by POSITION and the original index
  // of every position -- what bs_owner_fetch_dev maps back to the caller's order
  const int32_t* rg_omega = nullptr;
  const int32_t* rg_prio = nullptr;
  const int32_t* rg_seeds = nullptr;  // committed seeds (original indices, ascending = commit order), device
  int32_t rg_nplanes = 0;
  int forge_mode = 0;  // bs_selftest_forge_next
  int audit = 0;       // bs_set_audit
  hipStream_t side = nullptr;  // second stream of the grower (validate3 beside the owner passes)
  hipEvent_t sev[2] = {nullptr, nullptr};
  // 2-D raster scratch (bs_raster.hip)
  bs::DevBuf rs_keys_in, rs_keys_out, rs_vals_in, rs_vals_out, rs_cnt, rs_img, rs_tmp;
  // cell-sorted order of the last grid build (vals_out): spatially coherent iteration for gathers
  int64_t order_n = 0;
  const int32_t* order_xyz = nullptr;
  // neighbour rows as cell-sorted POSITIONS (rows in position order), written by the kNN kernels of the fused
  // pipeline for the same cloud / k / neigh buffer: the grower takes them instead of translating indices
  bs::DevBuf seg_npos;
  // scratch of the sharded pass (bs_sharded.hip): named by use there
  bs::DevBuf sh[25];  // (24 scratch buffers of bs_sharded.hip + the look-up table of bs_remap_rows_dev)
  std::vector<int32_t> sh_seeds;  // all committed seeds of the last bs_segment_sharded (global indices, ascending)
  int64_t sh_nloc = 0;            // points this rank grew
  bool sh_valid = false;
  const int32_t* npos_neigh = nullptr;
  const double* npos_normals = nullptr;
  int npos_k = 0;
};

namespace bs {

int fail(bs_ctx* ctx, int status, const char* what, hipError_t e = hipSuccess);

#define BS_HIP(ctx, call)                                     \
  do {                                                        \
    hipError_t _e = (call);                                   \
    if (_e != hipSuccess)                                     \
      return bs::fail((ctx), BS_ERR_HIP, #call, _e);          \
  } while (0)

// grid.hip
int build_grid(bs_ctx* ctx, const int32_t* d_xyz, const int32_t* d_gidx, int64_t n, double radius,
               int k, int cell_hint, GridDev* out);
int build_spatial_order(bs_ctx* ctx, const int32_t* d_xyz, int64_t n);
int bbox_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, int32_t bb[6]);
// knn.hip
int launch_knn_normals(bs_ctx* ctx, const GridDev& g, int64_t q_begin, int64_t q_end,
                       const bs_params& p, int32_t* d_neigh, double* d_normals, double cert_radius,
                       int64_t* n_uncertified, int32_t* d_npos = nullptr);
// grow.hip
int launch_region_grow_seq(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals,
                           const int32_t* d_neigh, int64_t n, const bs_params& p,
                           int32_t* d_plane_idx);
// grow_spec.hip
int launch_region_grow_spec(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals,
                            const int32_t* d_neigh, int64_t n, const bs_params& p,
                            int32_t* d_plane_idx);

}  // namespace bs
