// bs_grow_spec.hip -- rg_mode 2: exact multi-plane region growing by ordered
// speculation (gfx950).  Product code.
//
// The reference (seg_plane::get_planes / Broad,
// /root/reference/tmc3/my_function.cpp:180-258) scans seeds in index order; a
// seed attempt either FAILS at depth 0 (fewer than K-1 neighbours accepted,
// :238-239 -- the accepted ones stay labelled: "orphans", quirk Q2) or grows a
// plane that is committed (> th_point_count entries, :199) or rolled back
// (:203-208).  Two facts make an exact parallel schedule possible:
//
//  (a) The depth-0 tests use the seed's own normal and position
//      (:187-190), so WHICH neighbours of seed i can pass is a static
//      per-point mask H(i).  A failing attempt at i labels exactly
//      {c in H(i) : c still free}.  With owner[p] := index of the attempt that
//      keeps p labelled, the orphan makers obey
//          owner[c] = min{ i : c in H(i), attempt i happens (owner[i] >= i) },
//      a monotone system over lower indices only -> solved by Jacobi passes
//      (orphan_pass_kernel) to its unique fixed point.
//  (b) A plane attempt (all K-1 neighbours free and in H) is rare.  The lowest
//      candidates are grown CONCURRENTLY, one wavefront each, against the
//      tentative owner array; points are claimed with atomicMin(seed) so the
//      sequentially-earlier plane always wins, and every assumption a plane
//      made about not-yet-final owners is logged.  After the round the owner
//      fixed point is recomputed with the finished planes inserted and each
//      plane is validated (its seed still qualifies, every accepted point was
//      free at its time, every logged "taken" point is still taken by a lower
//      attempt).  Everything below the first invalid or newly appearing
//      candidate is FINAL -- identical to the sequential execution by
//      induction over the seed index.  The lowest candidate of a round always
//      validates, so every round makes progress.
//
// Final labels: plane_idx[p] = 1 + #(committed planes with seed < owner[p])
// (cur_planeId only advances on commit, :199-202), -1 if owner[p] is none.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bs_common.h"

namespace bs {

namespace {

constexpr int32_t INF = 0x7fffffff;
constexpr int MAX_WAVES = 1024;  // plane attempts grown concurrently per round

enum : int32_t { ST_NONE = 0, ST_DONE = 1, ST_FAILED0 = 2, ST_NOMEM = 3, ST_WATCHDOG = 4, ST_STOLEN = 5 };

struct SpecArgs {
  const int32_t* xyz;
  const double* normals;
  const int32_t* neigh;
  int64_t n;
  int K;
  double th;
  double cos_th;
  int64_t th_count;
  int32_t F;  // every attempt < F is final
  int32_t vec;  // neighbour rows are 16-byte aligned: int4 row loads/stores
};

struct PlaneOut {
  double normal[3];
  int64_t list_off;  // into the round pool
  int64_t list_n;
  int64_t log_off;
  int64_t log_n;
  int64_t steps;
  int32_t center[3];
  int32_t seed;
  int32_t status;
  int32_t keep;        // committed (list_n > th_count)
  int32_t consistent;  // set by validate2_kernel
  int32_t pad;
};

struct Pool {
  int32_t* base;
  unsigned long long* top;
  unsigned long long cap;
};

struct Slab {
  int64_t off;
  int64_t cap;
};

__device__ inline int ld_i32(const int32_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline int readlane_i32(int v, int lane)
{
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}

__device__ inline double readlane_f64(double v, int lane)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = readlane_i32(lo, lane);
  hi = readlane_i32(hi, lane);
  return __hiloint2double(hi, lo);
}

// wave-cooperative "realloc": make room for need entries, keeping used ones
__device__ inline bool slab_ensure(const Pool& pool, Slab& s, int64_t used, int64_t need, int lane)
{
  if (need <= s.cap)
    return true;
  int64_t ncap = s.cap * 2;
  if (ncap < need)
    ncap = need;
  if (ncap < 2048)
    ncap = 2048;
  ncap = (ncap + 3) & ~(int64_t)3;  // keep slab offsets 16-byte aligned (int4 stack slots)
  unsigned long long off = 0;
  if (lane == 0)
    off = atomicAdd(pool.top, (unsigned long long)ncap);
  off = ((unsigned long long)(uint32_t)readlane_i32((int)(off >> 32), 0) << 32) |
        (uint32_t)readlane_i32((int)(off & 0xffffffffu), 0);
  if (off + (unsigned long long)ncap > pool.cap)
    return false;
  for (int64_t t = lane; t < used; t += 64)
    pool.base[off + t] = ld_i32(pool.base + s.off + t);
  s.off = (int64_t)off;
  s.cap = ncap;
  return true;
}

// ---- (a) static depth-0 mask ------------------------------------------------
__global__ void static_mask_kernel(SpecArgs a, uint32_t* __restrict__ hmask)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= a.n)
    return;
  const double cnx = a.normals[3 * i], cny = a.normals[3 * i + 1], cnz = a.normals[3 * i + 2];
  const int ccx = a.xyz[3 * i], ccy = a.xyz[3 * i + 1], ccz = a.xyz[3 * i + 2];
  uint32_t m = 0;
  for (int t = 1; t < a.K; t++) {
    const int64_t c = a.neigh[i * a.K + t];
    const int dx = (int)((uint32_t)a.xyz[3 * c] - (uint32_t)ccx);
    const int dy = (int)((uint32_t)a.xyz[3 * c + 1] - (uint32_t)ccy);
    const int dz = (int)((uint32_t)a.xyz[3 * c + 2] - (uint32_t)ccz);
    const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
    const double dt = cnx * a.normals[3 * c] + cny * a.normals[3 * c + 1] + cnz * a.normals[3 * c + 2];
    if (dist <= a.th && dt >= a.cos_th)
      m |= 1u << (t - 1);
  }
  hmask[i] = m;
}

// ---- orphan-maker fixed point ------------------------------------------------
__global__ void orphan_pass_kernel(const uint32_t* __restrict__ hmask, const int32_t* __restrict__ neigh, int K,
                                   int64_t n, int32_t F, const uint8_t* __restrict__ ps,
                                   const int32_t* __restrict__ prev, int32_t* __restrict__ next)
{
  const int64_t i = F + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  uint32_t m = hmask[i];
  if (m == 0 || ps[i] || prev[i] < (int32_t)i)
    return;  // attempt i does not happen (already kept by a lower attempt) or is a plane handled apart
  const int32_t* row = neigh + i * K;
  while (m) {
    const int t = __ffs(m) - 1;
    m &= m - 1;
    atomicMin(&next[row[t + 1]], (int32_t)i);
  }
}

__global__ void diff_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t n, int* changed)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const bool d = i < n && a[i] != b[i];
  if (__ballot(d) && (threadIdx.x & 63) == 0)
    *changed = 1;
}

// ---- plane-attempt candidates -------------------------------------------------
__global__ void cand_flag_kernel(const uint32_t* __restrict__ hmask, const int32_t* __restrict__ neigh, int K,
                                 int64_t n, int32_t F, const uint8_t* __restrict__ ps,
                                 const int32_t* __restrict__ omega, uint8_t* __restrict__ flags, int32_t* min_idx)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  bool c = false;
  const uint32_t full = (K - 1 >= 32) ? 0xffffffffu : ((1u << (K - 1)) - 1u);
  if (i >= F && !ps[i] && hmask[i] == full && omega[i] >= (int32_t)i) {
    c = true;
    const int32_t* row = neigh + i * K;
    for (int t = 1; t < K; t++)
      c = c && omega[row[t]] >= (int32_t)i;
  }
  if (flags)
    flags[i] = c ? 1 : 0;
  if (c && min_idx)
    atomicMin(min_idx, (int32_t)i);
}

// ---- (b) speculative plane growth: one wavefront per candidate seed -------------
// Step engine: every lane that gathers a neighbour also prefetches that
// neighbour's own neighbour row, so the next Broad() call (first accepted
// child) starts from registers, and the other accepted children are pushed on
// the LIFO together with their rows (slot 0 of a row is never read by Broad,
// :224, so it carries the point id).  One dependent HBM round trip per step
// instead of three (pop -> row -> gather).
template <int KC>
__global__ __launch_bounds__(64) void grow_spec_kernel(SpecArgs a, const int32_t* __restrict__ cand, int ncand,
                                                       const int32_t* __restrict__ omega, int32_t* tag, Pool pool,
                                                       PlaneOut* __restrict__ out, int64_t step_cap)
{
  const int w = blockIdx.x;
  if (w >= ncand)
    return;
  const int lane = threadIdx.x;
  const int K = a.K, nc = K - 1;
  const bool vec = a.vec != 0;
  const int32_t seed = cand[w];
  Slab list = {0, 0}, stack = {0, 0}, log = {0, 0};
  int64_t ln = 1, sp = 0, logn = 0, steps = 0;
  int status = ST_DONE;
  double cnx = a.normals[3 * (int64_t)seed], cny = a.normals[3 * (int64_t)seed + 1],
         cnz = a.normals[3 * (int64_t)seed + 2];
  int ccx = a.xyz[3 * (int64_t)seed], ccy = a.xyz[3 * (int64_t)seed + 1], ccz = a.xyz[3 * (int64_t)seed + 2];
  double Sx = 0.0 + cnx, Sy = 0.0 + cny, Sz = 0.0 + cnz;
  uint32_t Cx = (uint32_t)ccx, Cy = (uint32_t)ccy, Cz = (uint32_t)ccz;
  if (!slab_ensure(pool, list, 0, 2048, lane) || !slab_ensure(pool, stack, 0, 256 * (int64_t)K, lane) ||
      !slab_ensure(pool, log, 0, 2048, lane)) {
    status = ST_NOMEM;
  } else {
    if (lane == 0)
      pool.base[list.off] = seed;
    const bool act = lane < nc;
    int cand_id = act ? a.neigh[(int64_t)seed * K + lane + 1] : 0;
    bool depth0 = true;
    for (;;) {
      if (++steps > step_cap) {
        status = ST_WATCHDOG;
        break;
      }
      int own = 0, tg = INF, px = 0, py = 0, pz = 0;
      double mx = 0, my = 0, mz = 0;
      int row[KC];
#pragma unroll
      for (int j = 0; j < KC; j++)
        row[j] = 0;
      if (act) {
        own = omega[cand_id];
        tg = ld_i32(tag + cand_id);
        px = a.xyz[3 * (int64_t)cand_id];
        py = a.xyz[3 * (int64_t)cand_id + 1];
        pz = a.xyz[3 * (int64_t)cand_id + 2];
        mx = a.normals[3 * (int64_t)cand_id];
        my = a.normals[3 * (int64_t)cand_id + 1];
        mz = a.normals[3 * (int64_t)cand_id + 2];
        const int32_t* rp = a.neigh + (int64_t)cand_id * K;
        if (vec) {
#pragma unroll
          for (int j = 0; j < KC; j += 4)
            if (j < K) {
              const int4 v = *reinterpret_cast<const int4*>(rp + j);
              row[j] = v.x;
              row[j + 1] = v.y;
              row[j + 2] = v.z;
              row[j + 3] = v.w;
            }
        } else {
#pragma unroll
          for (int j = 1; j < KC; j++)
            if (j < K)
              row[j] = rp[j];
        }
      }
      bool geo = false;
      if (act && tg != seed) {  // tg == seed: already labelled by this plane
        const int dx = (int)((uint32_t)px - (uint32_t)ccx);
        const int dy = (int)((uint32_t)py - (uint32_t)ccy);
        const int dz = (int)((uint32_t)pz - (uint32_t)ccz);
        const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
        const double dt = cnx * mx + cny * my + cnz * mz;
        geo = dist <= a.th && dt >= a.cos_th;
      }
      // taken by a sequentially earlier attempt?  (final, tentative, or in flight)
      const bool taken = geo && (own < seed || tg < seed);
      bool ok = geo && !taken;
      if (ok) {
        const int old = atomicMin(tag + cand_id, seed);
        if (old < seed)
          ok = false;  // lost the race to an earlier plane: now taken
      }
      const bool assume = geo && !ok && !(own < a.F);  // relied on a non-final owner: log it
      const unsigned long long am = __ballot(ok);
      const unsigned long long lm = __ballot(assume);
      const int cnt = __popcll(am);
      const int lcnt = __popcll(lm);
      if (lcnt) {
        if (!slab_ensure(pool, log, logn, logn + lcnt, lane)) {
          status = ST_NOMEM;
          break;
        }
        if (assume)
          pool.base[log.off + logn + __popcll(lm & ((1ull << lane) - 1ull))] = cand_id;
        logn += lcnt;
      }
      if (depth0 && cnt < nc) {
        status = ST_FAILED0;  // under speculation this seed is (currently) an orphan maker
        break;
      }
      depth0 = false;
      if (cnt) {
        if (!slab_ensure(pool, list, ln, ln + cnt, lane) ||
            !slab_ensure(pool, stack, sp * K, (sp + cnt) * K, lane)) {
          status = ST_NOMEM;
          break;
        }
        const int rank = __popcll(am & ((1ull << lane) - 1ull));
        if (ok)
          pool.base[list.off + ln + rank] = cand_id;
        unsigned long long mm = am;
        while (mm) {
          const int l = __ffsll(mm) - 1;
          mm &= mm - 1;
          Sx += readlane_f64(mx, l);
          Sy += readlane_f64(my, l);
          Sz += readlane_f64(mz, l);
          Cx += (uint32_t)readlane_i32(px, l);
          Cy += (uint32_t)readlane_i32(py, l);
          Cz += (uint32_t)readlane_i32(pz, l);
        }
        ln += cnt;
        const double nrm = __builtin_sqrt((Sx * Sx) + (Sy * Sy) + (Sz * Sz));
        cnx = Sx / nrm;
        cny = Sy / nrm;
        cnz = Sz / nrm;
        const uint64_t dn = (uint64_t)ln;
        ccx = (int32_t)((uint64_t)(int64_t)(int32_t)Cx / dn);
        ccy = (int32_t)((uint64_t)(int64_t)(int32_t)Cy / dn);
        ccz = (int32_t)((uint64_t)(int64_t)(int32_t)Cz / dn);
        // children 2..cnt go on the LIFO (reversed) with their rows; id in slot 0
        if (ok && rank > 0) {
          int32_t* slot = pool.base + stack.off + (sp + (cnt - 1 - rank)) * K;
          row[0] = cand_id;
          if (vec) {
#pragma unroll
            for (int j = 0; j < KC; j += 4)
              if (j < K)
                *reinterpret_cast<int4*>(slot + j) = make_int4(row[j], row[j + 1], row[j + 2], row[j + 3]);
          } else {
#pragma unroll
            for (int j = 0; j < KC; j++)
              if (j < K)
                slot[j] = row[j];
          }
        }
        sp += cnt - 1;
        // first child continues from registers: lane l takes row_f[l + 1]
        const int f = __ffsll(am) - 1;
        int nxt = 0;
#pragma unroll
        for (int j = 1; j < KC; j++) {
          const int t = readlane_i32(row[j], f);
          nxt = (lane == j - 1) ? t : nxt;
        }
        cand_id = nxt;
      } else {
        if (sp == 0)
          break;
        sp--;
        cand_id = act ? ld_i32(pool.base + stack.off + sp * K + lane + 1) : 0;
      }
    }
  }
  if (lane == 0) {
    PlaneOut o;
    o.normal[0] = cnx;
    o.normal[1] = cny;
    o.normal[2] = cnz;
    o.center[0] = ccx;
    o.center[1] = ccy;
    o.center[2] = ccz;
    o.list_off = list.off;
    o.list_n = ln;
    o.log_off = log.off;
    o.log_n = logn;
    o.steps = steps;
    o.seed = seed;
    o.status = status;
    o.keep = (status == ST_DONE && ln > a.th_count) ? 1 : 0;
    o.consistent = 0;
    o.pad = 0;
    out[w] = o;
  }
}

// every accepted point must still carry this plane's claim
__global__ __launch_bounds__(64) void validate1_kernel(PlaneOut* out, int ncand, const int32_t* __restrict__ pool,
                                                       const int32_t* __restrict__ tag, uint8_t* ps)
{
  const int w = blockIdx.x;
  if (w >= ncand || out[w].status != ST_DONE)
    return;
  const PlaneOut o = out[w];
  bool bad = false;
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += 64)
    bad = bad || tag[pool[o.list_off + t]] != o.seed;
  if (__ballot(bad)) {
    if (threadIdx.x == 0)
      out[w].status = ST_STOLEN;
  } else if (threadIdx.x == 0) {
    ps[o.seed] = 1;
  }
}

// kept points of finished, committed planes enter the base owner array
__global__ __launch_bounds__(64) void insert_kernel(const PlaneOut* __restrict__ out, int ncand,
                                                    const int32_t* __restrict__ pool, int32_t* base)
{
  const int w = blockIdx.x;
  if (w >= ncand)
    return;
  const PlaneOut o = out[w];
  if (o.status != ST_DONE || !o.keep)
    return;
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += 64)
    atomicMin(&base[pool[o.list_off + t]], o.seed);
}

__global__ __launch_bounds__(64) void validate2_kernel(PlaneOut* out, int ncand, const int32_t* __restrict__ pool,
                                                       const int32_t* __restrict__ omega,
                                                       const int32_t* __restrict__ neigh, int K)
{
  const int w = blockIdx.x;
  if (w >= ncand || out[w].status != ST_DONE)
    return;
  const PlaneOut o = out[w];
  const int32_t s = o.seed;
  bool bad = false;
  if (threadIdx.x == 0)
    bad = omega[s] < s;  // (1) the seed is still free at its time ...
  if (threadIdx.x >= 1 && threadIdx.x < K)
    bad = omega[neigh[(int64_t)s * K + threadIdx.x]] < s;  // ... and so are its K-1 neighbours
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += 64)  // (2) accepted points were free
    bad = bad || omega[pool[o.list_off + t]] < s;
  for (int64_t t = threadIdx.x; t < o.log_n; t += 64)  // (3) assumed-taken points are taken
    bad = bad || !(omega[pool[o.log_off + t]] < s);
  const unsigned long long b = __ballot(bad);
  if (threadIdx.x == 0)
    out[w].consistent = b ? 0 : 1;
}

__global__ void finalize_owner_kernel(const int32_t* __restrict__ omega, int32_t first_bad, int64_t n,
                                      int32_t* __restrict__ owner_final)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t o = omega[i];
  owner_final[i] = o < first_bad ? o : INF;
}

__global__ void copy_list_kernel(const int32_t* __restrict__ pool, int64_t src, int64_t cnt, int32_t* dst)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < cnt)
    dst[i] = pool[src + i];
}

__global__ void label_kernel(const int32_t* __restrict__ owner, int64_t n, const int32_t* __restrict__ seeds,
                             int np, int32_t* __restrict__ plane_idx)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t o = owner[i];
  if (o == INF) {
    plane_idx[i] = -1;
    return;
  }
  int lo = 0, hi = np;  // number of committed seeds < o
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seeds[mid] < o)
      lo = mid + 1;
    else
      hi = mid;
  }
  plane_idx[i] = 1 + lo;
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = v;
}

inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

}  // namespace

// Jacobi iteration of the orphan-maker system to its fixed point.
// base: owners that are given (finals + inserted planes); result in *result.
static int orphan_fixpoint(bs_ctx* ctx, const uint32_t* hmask, const int32_t* neigh, int K, int64_t n, int32_t F,
                           const uint8_t* ps, const int32_t* base, int32_t* bufA, int32_t* bufB, int* d_changed,
                           int32_t** result, int64_t* passes)
{
  hipStream_t st = ctx->stream;
  int32_t* prev = bufA;
  int32_t* next = bufB;
  BS_HIP(ctx, hipMemcpyAsync(prev, base, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
  const int64_t m = n - F;
  for (int it = 0; it < 100000; it++) {
    BS_HIP(ctx, hipMemcpyAsync(next, base, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
    if (m > 0)
      orphan_pass_kernel<<<nblk(m, 256), 256, 0, st>>>(hmask, neigh, K, n, F, ps, prev, next);
    BS_HIP(ctx, hipMemsetAsync(d_changed, 0, sizeof(int), st));
    diff_kernel<<<nblk(n, 256), 256, 0, st>>>(prev, next, n, d_changed);
    int ch = 0;
    BS_HIP(ctx, hipMemcpyAsync(&ch, d_changed, sizeof ch, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    (*passes)++;
    std::swap(prev, next);
    if (!ch) {
      *result = prev;
      return BS_OK;
    }
  }
  return fail(ctx, BS_ERR_INTERNAL, "orphan fixed point did not converge");
}

int launch_region_grow_spec(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals, const int32_t* d_neigh,
                            int64_t n, const bs_params& p, int32_t* d_plane_idx)
{
  hipStream_t st = ctx->stream;
  ctx->rg_valid = false;
  const int K = p.k;
  const int64_t list_cap = 2 * n + 64;
  const int64_t planes_cap = n / std::max(1, p.th_point_count) + 64;
  // round pool: lists + stacks + logs of every concurrent attempt
  const unsigned long long pool_cap =
      (unsigned long long)std::max<int64_t>(std::min<int64_t>(64 * n, (int64_t)3 << 30), 8 * n + (int64_t)(2 * 2048 + 256 * K) * 2 * MAX_WAVES);
  BS_HIP(ctx, ctx->rg_list.reserve(sizeof(int32_t) * list_cap));
  BS_HIP(ctx, ctx->rg_planes.reserve(sizeof(PlaneRec) * planes_cap));
  BS_HIP(ctx, ctx->rg_stats.reserve(sizeof(GrowStats)));
  // aux layout (int32 units): misc[1024] | hmask | owner_final | bufA | bufB | base | tag | cand | seeds |
  //                            flags(u8) | ps(u8) | PlaneOut[MAX_WAVES]
  const size_t aux_bytes = sizeof(int32_t) * (size_t)(7 * n + planes_cap + 1024 + 64) + (size_t)2 * n + 4096 +
                           sizeof(PlaneOut) * MAX_WAVES;
  BS_HIP(ctx, ctx->rg_aux.reserve(aux_bytes));
  BS_HIP(ctx, ctx->rg_stack.reserve(sizeof(int32_t) * pool_cap));
  int32_t* aux = ctx->rg_aux.as<int32_t>();
  int32_t* d_misc = aux;  // [0]=changed [1]=ncand [2]=min_idx ; [16..17] pool top (u64)
  uint32_t* hmask = (uint32_t*)(aux + 1024);
  int32_t* owner_final = aux + 1024 + n;
  int32_t* bufA = aux + 1024 + 2 * n;
  int32_t* bufB = aux + 1024 + 3 * n;
  int32_t* base = aux + 1024 + 4 * n;
  int32_t* tag = aux + 1024 + 5 * n;
  int32_t* d_cand = aux + 1024 + 6 * n;  // select output (n entries)
  int32_t* d_seeds = aux + 1024 + 7 * n;  // committed seeds (planes_cap)
  uint8_t* flags = (uint8_t*)(aux + 1024 + 7 * n + planes_cap + 64);
  uint8_t* ps = flags + n;
  PlaneOut* d_out = (PlaneOut*)(((uintptr_t)(ps + n) + 255) & ~(uintptr_t)255);
  unsigned long long* d_pool_top = (unsigned long long*)(d_misc + 16);
  Pool pool = {ctx->rg_stack.as<int32_t>(), d_pool_top, pool_cap};

  SpecArgs a;
  a.xyz = d_xyz;
  a.normals = d_normals;
  a.neigh = d_neigh;
  a.n = n;
  a.K = K;
  a.th = (double)p.th_thickness;
  a.cos_th = p.cos_th;
  a.th_count = p.th_point_count;
  a.F = 0;
  a.vec = ((K & 3) == 0 && ((uintptr_t)d_neigh & 15) == 0) ? 1 : 0;

  static_mask_kernel<<<nblk(n, 256), 256, 0, st>>>(a, hmask);
  fill_i32_kernel<<<nblk(n, 256), 256, 0, st>>>(owner_final, n, INF);
  BS_HIP(ctx, hipMemsetAsync(ps, 0, n, st));

  std::vector<PlaneRec> recs;
  std::vector<int32_t> seeds;
  std::vector<PlaneOut> h_out(MAX_WAVES);
  int64_t list_used = 0, largest = 0, attempts = 0, rounds = 0, passes = 0;
  int32_t F = 0;
  size_t sel_tmp = 0;
  {
    hipcub::CountingInputIterator<int32_t> it(0);
    BS_HIP(ctx, hipcub::DeviceSelect::Flagged(nullptr, sel_tmp, it, flags, d_cand, d_misc + 1, (int)n, st));
    BS_HIP(ctx, ctx->cub_tmp.reserve(sel_tmp));
  }
  int max_waves = MAX_WAVES;
  for (;;) {
    rounds++;
    a.F = F;
    // tentative owners with every open attempt treated as an orphan maker
    int32_t* omega = nullptr;
    int rc = orphan_fixpoint(ctx, hmask, d_neigh, K, n, F, ps, owner_final, bufA, bufB, d_misc, &omega, &passes);
    if (rc != BS_OK)
      return rc;
    // lowest plane-attempt candidates
    cand_flag_kernel<<<nblk(n, 256), 256, 0, st>>>(hmask, d_neigh, K, n, F, ps, omega, flags, nullptr);
    {
      hipcub::CountingInputIterator<int32_t> it(0);
      size_t tb = sel_tmp;
      BS_HIP(ctx, hipcub::DeviceSelect::Flagged(ctx->cub_tmp.p, tb, it, flags, d_cand, d_misc + 1, (int)n, st));
    }
    int32_t ncand_all = 0;
    BS_HIP(ctx, hipMemcpyAsync(&ncand_all, d_misc + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    if (ncand_all == 0) {
      // no plane attempt left: the fixed point is the final owner array
      BS_HIP(ctx, hipMemcpyAsync(owner_final, omega, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
      break;
    }
    const int ncand = std::min<int>(ncand_all, max_waves);
    // grow them concurrently
    fill_i32_kernel<<<nblk(n, 256), 256, 0, st>>>(tag, n, INF);
    BS_HIP(ctx, hipMemsetAsync(d_pool_top, 0, sizeof(unsigned long long), st));
    if (K <= 16)
      grow_spec_kernel<16><<<ncand, 64, 0, st>>>(a, d_cand, ncand, omega, tag, pool, d_out, 512 * n + 4096);
    else
      grow_spec_kernel<32><<<ncand, 64, 0, st>>>(a, d_cand, ncand, omega, tag, pool, d_out, 512 * n + 4096);
    validate1_kernel<<<ncand, 64, 0, st>>>(d_out, ncand, pool.base, tag, ps);
    // owner base with the finished planes inserted, then the new fixed point
    BS_HIP(ctx, hipMemcpyAsync(base, owner_final, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
    insert_kernel<<<ncand, 64, 0, st>>>(d_out, ncand, pool.base, base);
    int32_t* omega2 = nullptr;
    // omega lives in bufA/bufB; the second fixed point reuses them, so keep nothing from the first
    rc = orphan_fixpoint(ctx, hmask, d_neigh, K, n, F, ps, base, bufA, bufB, d_misc, &omega2, &passes);
    if (rc != BS_OK)
      return rc;
    validate2_kernel<<<ncand, 64, 0, st>>>(d_out, ncand, pool.base, omega2, d_neigh, K);
    int32_t inf = INF;
    BS_HIP(ctx, hipMemcpyAsync(d_misc + 2, &inf, sizeof inf, hipMemcpyHostToDevice, st));
    cand_flag_kernel<<<nblk(n, 256), 256, 0, st>>>(hmask, d_neigh, K, n, F, ps, omega2, nullptr, d_misc + 2);
    int32_t new_min = INF;
    BS_HIP(ctx, hipMemcpyAsync(&new_min, d_misc + 2, sizeof new_min, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipMemcpyAsync(h_out.data(), d_out, sizeof(PlaneOut) * ncand, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    // first attempt (by seed index) whose result is not established
    int32_t first_bad = new_min;
    bool nomem_lowest = false;
    for (int w = 0; w < ncand; w++) {
      const PlaneOut& o = h_out[w];
      if (o.status == ST_WATCHDOG)
        return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): watchdog");
      if (o.status == ST_DONE && !o.consistent)
        first_bad = std::min(first_bad, o.seed);
      if (w == 0 && o.status == ST_NOMEM)
        nomem_lowest = true;
    }
    if (nomem_lowest) {
      if (max_waves == 1)
        return fail(ctx, BS_ERR_NOMEM, "region grow (speculative): round pool exhausted");
      max_waves = std::max(1, max_waves / 8);
    }
    if (getenv("BS_DEBUG")) {
      int cnt[6] = {0, 0, 0, 0, 0, 0}, cons = 0;
      int64_t maxsteps = 0, sumsteps = 0, maxlist = 0;
      for (int w = 0; w < ncand; w++) {
        cnt[h_out[w].status]++;
        cons += h_out[w].status == ST_DONE && h_out[w].consistent;
        maxsteps = std::max(maxsteps, h_out[w].steps);
        sumsteps += h_out[w].steps;
        maxlist = std::max(maxlist, h_out[w].list_n);
      }
      fprintf(stderr,
              "[bs] round %ld F=%d ncand_all=%d grown=%d done=%d (consistent %d) failed0=%d nomem=%d stolen=%d "
              "first_bad=%d new_min=%d lowest=%d highest=%d maxsteps=%ld sumsteps=%ld maxlist=%ld passes=%ld\n",
              (long)rounds, F, ncand_all, ncand, cnt[ST_DONE], cons, cnt[ST_FAILED0], cnt[ST_NOMEM], cnt[ST_STOLEN],
              first_bad, new_min, h_out[0].seed, h_out[ncand - 1].seed, (long)maxsteps, (long)sumsteps,
              (long)maxlist, (long)passes);
    }
    // commit everything below first_bad, in seed order (cand is sorted)
    int finals = 0;
    for (int w = 0; w < ncand; w++) {
      const PlaneOut& o = h_out[w];
      if (o.seed >= first_bad)
        break;
      if (o.status != ST_DONE || !o.consistent)
        continue;  // an orphan maker under the final owners (seed or a neighbour was taken)
      finals++;
      attempts++;
      if (!o.keep)
        continue;  // rolled back: no trace
      if (list_used + o.list_n > list_cap || (int64_t)recs.size() >= planes_cap)
        return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): list pool overflow");
      copy_list_kernel<<<nblk(o.list_n, 256), 256, 0, st>>>(pool.base, o.list_off, o.list_n,
                                                            ctx->rg_list.as<int32_t>() + list_used);
      PlaneRec r;
      for (int c = 0; c < 3; c++) {
        r.normal[c] = o.normal[c];
        r.center[c] = o.center[c];
      }
      r.list_off = list_used;
      r.list_n = o.list_n;
      r.id = (int32_t)recs.size() + 1;
      r.seed = o.seed;
      r.pad = 0;
      recs.push_back(r);
      seeds.push_back(o.seed);
      list_used += o.list_n;
      largest = std::max<int64_t>(largest, o.list_n);
    }
    if (first_bad == INF) {
      // no open plane attempt is left (an ungrown candidate would have shown up
      // in new_min): every remaining attempt is an orphan maker and omega2 is
      // their fixed point
      BS_HIP(ctx, hipMemcpyAsync(owner_final, omega2, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
      break;
    }
    finalize_owner_kernel<<<nblk(n, 256), 256, 0, st>>>(omega2, first_bad, n, owner_final);
    F = first_bad;
    BS_HIP(ctx, hipMemsetAsync(ps, 0, n, st));
    if (finals == 0 && !nomem_lowest && first_bad <= h_out[0].seed)
      return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): no progress");
    if (rounds > 4 * n + 16)
      return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): round limit");
  }
  // labels and plane records
  const int np = (int)recs.size();
  if (np > 0) {
    BS_HIP(ctx, hipMemcpyAsync(d_seeds, seeds.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, st));
    BS_HIP(ctx, hipMemcpyAsync(ctx->rg_planes.p, recs.data(), sizeof(PlaneRec) * np, hipMemcpyHostToDevice, st));
  }
  label_kernel<<<nblk(n, 256), 256, 0, st>>>(owner_final, n, d_seeds, np, d_plane_idx);
  GrowStats hs;
  hs.n_planes = np;
  hs.error = 0;
  hs.list_used = list_used;
  hs.seed_attempts = attempts;
  hs.largest = largest;
  hs.steps = passes;
  BS_HIP(ctx, hipMemcpyAsync(ctx->rg_stats.p, &hs, sizeof hs, hipMemcpyHostToDevice, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  ctx->rg_n = n;
  ctx->rg_valid = true;
  ctx->tm.largest_plane = largest;
  ctx->tm.n_seed_attempts = attempts;
  ctx->tm.rg_rounds = rounds;
  return BS_OK;
}

}  // namespace bs
