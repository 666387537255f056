// bs_shard.hip -- device building blocks of the multi-GPU path (gfx950).  Product code.
//
// Stage 3 of the reference (seg_plane::get_planes / Broad, /root/reference/tmc3/my_function.cpp:180-258) is a
// scan over ALL seeds in index order, but information only ever travels along kNN edges: Broad(u) tests and
// labels neigh[u][1..K-1] (:224-233), a failed seed labels a subset of its own row (:238-239).  Two points in
// different connected components of the (undirected) kNN graph therefore never influence each other, and the
// only thing the components share is cur_planeId, which advances once per committed plane (:199-202):
//     planeIdx[p] = 1 + #(committed seeds < owner[p])          owner[p] = the seed attempt that left p labelled.
// So stage 3 shards EXACTLY by components: every rank grows whole components with the single-GPU scheduler on a
// local cloud whose index order is the global one restricted to it, the committed seeds are all-gathered, and the
// labels follow from the owners.  This file holds what that needs on the device:
//   bs_cc_hook_dev            connected components of a distributed edge set by hooking on a global parent array
//                             (the "label union-find" whose all-reduce(MIN) north_star asks for),
//   bs_owner_fetch_dev        owner[] of the last speculative grow in the caller's index order,
//   bs_labels_from_owner_dev  planeIdx from owners + the sorted global list of committed seeds,
//   bs_remap_rows_dev         neighbour rows from global indices to local ones (binary search in the sorted
//                             global indices of the local cloud).
#include <algorithm>

#include <cstdlib>
#include "bs_common.h"

namespace bs {

namespace {

// Union-find on a parent array over GLOBAL point ids; invariant parent[x] <= x, roots point at themselves, a
// pointer only ever moves to a smaller member of the same component (so a stale read is still an ancestor).
// Loads go to L2 (agent scope): CUs do not see each other's stores through their L1.
__device__ inline int32_t uf_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline int32_t uf_find(int32_t* parent, int32_t x)
{
  int32_t p = uf_load(parent + x);
  while (p != x) {
    const int32_t gp = uf_load(parent + p);
    if (gp != p)  // path halving; racing writers only ever store smaller ancestors
      __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = p;
    p = gp;
  }
  return x;
}

// hook the larger root under the smaller one (ECL-CC style); returns true iff THIS call performed a union
__device__ inline bool uf_union(int32_t* parent, int32_t u, int32_t v)
{
  int32_t ru = uf_find(parent, u), rv = uf_find(parent, v);
  while (ru != rv) {
    const int32_t hi = ru > rv ? ru : rv, lo = ru > rv ? rv : ru;
    const int32_t old = atomicCAS(parent + hi, hi, lo);
    if (old == hi)
      return true;
    // hi stopped being a root in the meantime: continue from where it points now
    ru = uf_find(parent, old);
    rv = lo;
  }
  return false;
}

__global__ __launch_bounds__(256) void cc_hook_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ gidx,
                                                      int64_t m, int k, int32_t* parent, unsigned long long* hooks)
{
  // one thread per (row, 4 slots): consecutive lanes read consecutive int4 of the row-major edge array
  const int kq = (k + 3) >> 2;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  unsigned int mine = 0;
  if (t < m * kq) {
    const int64_t i = t / kq;
    const int q = (int)(t - i * kq);
    const int32_t u = gidx ? gidx[i] : (int32_t)i;
    for (int j = 4 * q; j < 4 * q + 4 && j < k; j++) {
      const int32_t v = rows[i * k + j];
      if (v != u && uf_union(parent, u, v))
        mine++;
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine)
    atomicAdd(hooks, (unsigned long long)mine);
}

// after the hooking launch the forest is static: every node this rank touched (own points and the targets of
// their rows) is pointed straight at its root, so that the all-reduce(MIN) of the parent arrays tells the owner
// of a foreign target what this rank learned about it
__global__ __launch_bounds__(256) void cc_compress_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ gidx,
                                                          int64_t m, int k, int32_t* parent)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= m * (int64_t)k)
    return;
  const int64_t i = t / k;
  const int j = (int)(t - i * k);
  const int32_t x = j == 0 ? (gidx ? gidx[i] : (int32_t)i) : rows[t];
  int32_t r = x, p = parent[r];
  while (p != r) {
    r = p;
    p = parent[r];
  }
  if (parent[x] != r)
    parent[x] = r;  // (racing writers store the same root)
  if (j == 0) {     // slot 0 is normally the point itself; with duplicate coordinates it may be another point
    const int32_t y = rows[t];
    if (y != x) {
      int32_t r2 = y, p2 = parent[r2];
      while (p2 != r2) {
        r2 = p2;
        p2 = parent[r2];
      }
      if (parent[y] != r2)
        parent[y] = r2;
    }
  }
}

__global__ void owner_fetch_kernel(const int32_t* __restrict__ omega, const int32_t* __restrict__ prio, int64_t n,
                                   int32_t* __restrict__ owner)
{
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= n)
    return;
  const int32_t o = omega[s];
  owner[prio[s]] = o == 0x7fffffff ? -1 : o;
}

__global__ void labels_from_owner_kernel(const int32_t* __restrict__ owner, int64_t n, const int32_t* __restrict__ seeds,
                                         int np, int32_t* __restrict__ labels)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t o = owner[i];
  if (o < 0) {
    labels[i] = -1;
    return;
  }
  int lo = 0, hi = np;  // number of committed seeds < o  (my_function.cpp:199-202)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seeds[mid] < o)
      lo = mid + 1;
    else
      hi = mid;
  }
  labels[i] = 1 + lo;
}

// the same through a look-up table over the global index range [0, span): table[g] = local index or -1
__global__ void remap_table_fill_kernel(const int32_t* __restrict__ sorted_gidx, int32_t n, int32_t* __restrict__ table)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    table[sorted_gidx[i]] = (int32_t)i;
}

__global__ void remap_rows_table_kernel(const int32_t* __restrict__ rows, int64_t total, const int32_t* __restrict__ table,
                                        int64_t span, int32_t* __restrict__ out, int* missing)
{
  bool miss = false;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t g = rows[t];
    const int32_t l = (g >= 0 && g < span) ? table[g] : -1;
    out[t] = l >= 0 ? l : 0;
    miss = miss || l < 0;
  }
  if (miss)
    atomicAdd(missing, 1);
}

__global__ void remap_rows_kernel(const int32_t* __restrict__ rows, int64_t total, const int32_t* __restrict__ sorted_gidx,
                                  int32_t n, int32_t* __restrict__ out, int* missing)
{
  bool miss = false;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t g = rows[t];
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted_gidx[mid] < g)
        lo = mid + 1;
      else
        hi = mid;
    }
    const bool found = lo < n && sorted_gidx[lo] == g;
    out[t] = found ? lo : 0;
    miss = miss || !found;
  }
  if (__syncthreads_or(miss) && threadIdx.x == 0)
    *missing = 1;
}

inline int blocks_for(int64_t n, int b) { return (int)std::min<int64_t>((n + b - 1) / b, 0x7fffffff); }

}  // namespace

}  // namespace bs

using namespace bs;

extern "C" {

int bs_cc_hook_dev(bs_ctx* ctx, const int32_t* d_rows, const int32_t* d_gidx, int64_t m, int32_t k, int32_t* d_parent,
                   int64_t n_total, int64_t* n_hooks)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_parent || (m > 0 && !d_rows) || m < 0 || k < 1 || k > 64 || n_total <= 0 || n_total >= (int64_t)INT32_MAX - 64)
    return fail(ctx, BS_ERR_INVALID, "bs_cc_hook_dev: bad arguments");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->misc.reserve(256));
  hipStream_t st = ctx->stream;
  unsigned long long* d_hooks = ctx->misc.as<unsigned long long>() + 10;
  BS_HIP(ctx, hipMemsetAsync(d_hooks, 0, sizeof(unsigned long long), st));
  if (m > 0) {
    const int kq = (k + 3) >> 2;
    cc_hook_kernel<<<blocks_for(m * kq, 256), 256, 0, st>>>(d_rows, d_gidx, m, k, d_parent, d_hooks);
    cc_compress_kernel<<<blocks_for(m * (int64_t)k, 256), 256, 0, st>>>(d_rows, d_gidx, m, k, d_parent);
  }
  unsigned long long h = 0;
  BS_HIP(ctx, hipMemcpyAsync(&h, d_hooks, sizeof h, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  if (n_hooks)
    *n_hooks = (int64_t)h;
  return BS_OK;
}

int bs_owner_fetch_dev(bs_ctx* ctx, int32_t* d_owner)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!d_owner)
    return fail(ctx, BS_ERR_INVALID, "null device pointer");
  if (!ctx->rg_valid || !ctx->rg_omega || !ctx->rg_prio)
    return fail(ctx, BS_ERR_INVALID, "no speculative region-grow result on this context (rg_mode 1 keeps no owners)");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  owner_fetch_kernel<<<blocks_for(ctx->rg_n, 256), 256, 0, ctx->stream>>>(ctx->rg_omega, ctx->rg_prio, ctx->rg_n, d_owner);
  BS_HIP(ctx, hipGetLastError());
  return BS_OK;
}

int bs_plane_seeds_dev(bs_ctx* ctx, int32_t* d_seeds, int64_t cap, int32_t* n_planes)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!ctx->rg_valid || (ctx->rg_nplanes > 0 && !ctx->rg_seeds))
    return fail(ctx, BS_ERR_INVALID, "no speculative region-grow result on this context");
  if (n_planes)
    *n_planes = ctx->rg_nplanes;
  const int64_t cnt = std::min<int64_t>(cap, ctx->rg_nplanes);
  if (d_seeds && cnt > 0) {
    BS_HIP(ctx, hipSetDevice(ctx->device));
    BS_HIP(ctx, hipMemcpyAsync(d_seeds, ctx->rg_seeds, sizeof(int32_t) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return BS_OK;
}

int bs_stream_sync(bs_ctx* ctx)
{
  if (!ctx)
    return BS_ERR_INVALID;
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BS_OK;
}

int bs_labels_from_owner_dev(bs_ctx* ctx, const int32_t* d_owner, int64_t n, const int32_t* d_seeds, int32_t n_seeds,
                             int32_t* d_plane_idx)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (n < 0 || n_seeds < 0 || (n > 0 && (!d_owner || !d_plane_idx)) || (n_seeds > 0 && !d_seeds))
    return fail(ctx, BS_ERR_INVALID, "bs_labels_from_owner_dev: bad arguments");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  if (n > 0)
    labels_from_owner_kernel<<<blocks_for(n, 256), 256, 0, ctx->stream>>>(d_owner, n, d_seeds, n_seeds, d_plane_idx);
  BS_HIP(ctx, hipGetLastError());
  return BS_OK;
}

int bs_remap_rows_dev(bs_ctx* ctx, const int32_t* d_rows, int64_t n_rows, int32_t k, const int32_t* d_sorted_gidx,
                      int64_t n, int32_t* d_out, int32_t* n_missing)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (n_rows < 0 || k < 1 || n < 0 || n >= (int64_t)INT32_MAX - 64 || (n_rows > 0 && (!d_rows || !d_out || !d_sorted_gidx)))
    return fail(ctx, BS_ERR_INVALID, "bs_remap_rows_dev: bad arguments");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  BS_HIP(ctx, ctx->misc.reserve(256));
  hipStream_t st = ctx->stream;
  int* d_miss = ctx->misc.as<int>() + 52;
  BS_HIP(ctx, hipMemsetAsync(d_miss, 0, sizeof(int), st));
  if (n_rows > 0) {
    // Many rows: one random access per entry into a table over [0, largest index] instead of a 20-25 step binary
    // search per entry (the table is n_total ints at most: 200 MB at 50 M points, Infinity-Cache sized).
    int32_t last = -1;
    if (n > 0 && n_rows * (int64_t)k >= (1 << 22) && !getenv("BS_REMAP_BSEARCH")) {
      BS_HIP(ctx, hipMemcpyAsync(&last, d_sorted_gidx + (n - 1), sizeof last, hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
    }
    const int blocks = (int)std::min<int64_t>((n_rows * k + 255) / 256, 1 << 20);
    if (last >= 0) {
      const int64_t span = (int64_t)last + 1;
      BS_HIP(ctx, ctx->sh[24].reserve(sizeof(int32_t) * (size_t)span));
      int32_t* table = ctx->sh[24].as<int32_t>();
      BS_HIP(ctx, hipMemsetAsync(table, 0xFF, sizeof(int32_t) * (size_t)span, st));
      remap_table_fill_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(d_sorted_gidx, (int32_t)n, table);
      remap_rows_table_kernel<<<blocks, 256, 0, st>>>(d_rows, n_rows * (int64_t)k, table, span, d_out, d_miss);
    } else {
      remap_rows_kernel<<<blocks, 256, 0, st>>>(d_rows, n_rows * (int64_t)k, d_sorted_gidx, (int32_t)n, d_out, d_miss);
    }
  }
  int miss = 0;
  BS_HIP(ctx, hipMemcpyAsync(&miss, d_miss, sizeof miss, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  if (n_missing)
    *n_missing = miss;
  return BS_OK;
}

}  // extern "C"
