// bs_knn.hip -- exact k-nearest-neighbour lists + PCA normals on the hashed
// grid (gfx950).  Product code.
//
// Replaces, for every query point,
//   KDTreeFlann::SearchKNN(p_i, K)                 my_function.h:71-78
//   EstimateNormals(KDTreeSearchParamHybrid(r,M))  my_function.h:63
//   OrientNormalsToAlignWithDirection((0,0,1))     my_function.h:64
// (paths relative to /root/reference/tmc3/).  One grid pass serves both
// searches: the k best (d^2, index) keys are kept sorted in registers while the
// nine moment sums of the d^2 < r^2 neighbourhood are accumulated as exact
// integers; the normal is the fused epilogue (bs_normal.h).
//
// Canonical neighbour order: ascending exact integer d^2, ties by ascending
// global index -- one 64-bit key (d^2 << 32 | index) per candidate.
//
// Exactness: after ring rho of cells around the query's cell every point
// outside the examined block is at distance >= R (distance to the block's
// faces).  The k-list is final once its k-th d^2 < R^2, the hybrid set once
// R^2 >= r^2.  Queries the fast kernel cannot certify within BS_FAST_RINGS
// rings, or whose r-ball holds more than max_nn points (needs a top-max_nn
// selection), are appended to a work list and finished by knn_general_kernel,
// which keeps explicit sorted lists in scratch and falls back to a full scan.
#include <cstdlib>

#include "bs_common.h"
#include "bs_normal.h"

namespace bs {

namespace {

constexpr int BS_FAST_RINGS = 2;     // 5x5x5 cells at most in the fast kernel
constexpr int BS_GENERAL_RINGS = 6;  // then a full scan

__device__ inline bool cell_lookup(const GridDev& g, uint32_t cx, uint32_t cy, uint32_t cz, int& start,
                                   int& end)
{
  const uint64_t k = pack_cell(cx, cy, cz);
  uint32_t h = hash_cell(k) & g.hmask;
  for (;;) {
    const int4 raw = *reinterpret_cast<const int4*>(&g.table[h]);
    const uint64_t ek = (uint64_t)(uint32_t)raw.x | ((uint64_t)(uint32_t)raw.y << 32);
    if (ek == k) {
      start = raw.z;
      end = raw.w;
      return true;
    }
    if (ek == ~0ull)
      return false;
    h = (h + 1) & g.hmask;
  }
}

// distance from the query to the first coordinate outside the (2rho+1)^3 block
__device__ inline bool guaranteed_radius(const GridDev& g, const int q[3], const int ci[3], int rho,
                                         uint64_t& R2)
{
  uint32_t R = 0xFFFFFFFFu;
  bool bounded = false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (ci[a] - rho > 0) {
      int lo = g.mn[a] + (ci[a] - rho) * g.cell;
      R = min(R, (uint32_t)(q[a] - lo + 1));
      bounded = true;
    }
    if (ci[a] + rho < g.dim[a] - 1) {
      int hi = g.mn[a] + (ci[a] + rho + 1) * g.cell - 1;
      R = min(R, (uint32_t)(hi + 1 - q[a]));
      bounded = true;
    }
  }
  R2 = (uint64_t)R * (uint64_t)R;
  return bounded;
}

__device__ inline void moments_add(Moments& m, int x, int y, int z)
{
  m.sx += x;
  m.sy += y;
  m.sz += z;
  m.sxx += (uint64_t)((int64_t)x * x);
  m.syy += (uint64_t)((int64_t)y * y);
  m.szz += (uint64_t)((int64_t)z * z);
  m.sxy += (int64_t)x * y;
  m.sxz += (int64_t)x * z;
  m.syz += (int64_t)y * z;
  m.n += 1;
}

template <int KC>
__global__ __launch_bounds__(256) void knn_fast_kernel(GridDev g, int64_t q_begin, int64_t q_end, int K,
                                                       int max_nn, double r2, int32_t* __restrict__ neigh,
                                                       double* __restrict__ normals,
                                                       int32_t* __restrict__ fb_list,
                                                       int32_t* __restrict__ fb_count, uint64_t cert_r2,
                                                       unsigned long long* __restrict__ uncert,
                                                       int32_t* __restrict__ npos, unsigned long long* __restrict__ tie_rows)
{
  const int64_t s = xcd_logical_block() * (int64_t)blockDim.x + threadIdx.x;
  if (s >= g.n)
    return;
  const int32_t loc = g.slocal[s];
  if (loc < q_begin || loc >= q_end)
    return;
  // runner-up: the smallest key that is NOT in the list (only needed when the list fills every slot, K == KC):
  // the (K+1)-th neighbour decides whether the list's boundary is an equal-d^2 tie
  uint64_t runner = ~0ull;
  const int4 P = g.spts[s];
  const int q[3] = {P.x, P.y, P.z};
  const int ci[3] = {(int)((uint32_t)(P.x - g.mn[0]) / (uint32_t)g.cell),
                     (int)((uint32_t)(P.y - g.mn[1]) / (uint32_t)g.cell),
                     (int)((uint32_t)(P.z - g.mn[2]) / (uint32_t)g.cell)};
  uint64_t best[KC];
  // cell-sorted POSITION of every kept candidate, carried through the insertion network as a payload:
  // the region grower addresses points by position (bs_grow_spec.hip) and would otherwise have to look
  // every neighbour up by its index -- one random HBM access per edge
  int bpos[KC];
#pragma unroll
  for (int j = 0; j < KC; j++) {
    best[j] = ~0ull;
    bpos[j] = 0;
  }
  Moments m = {};
  bool done = false;
  for (int rho = 0; rho <= BS_FAST_RINGS && !done; rho++) {
    for (int dz = -rho; dz <= rho; dz++) {
      const int cz = ci[2] + dz;
      if (cz < 0 || cz >= g.dim[2])
        continue;
      for (int dy = -rho; dy <= rho; dy++) {
        const int cy = ci[1] + dy;
        if (cy < 0 || cy >= g.dim[1])
          continue;
        const bool face = (dz == -rho || dz == rho || dy == -rho || dy == rho);
        const int step = face ? 1 : (rho > 0 ? 2 * rho : 1);
        for (int dx = -rho; dx <= rho; dx += step) {
          const int cx = ci[0] + dx;
          if (cx < 0 || cx >= g.dim[0])
            continue;
          int cs, ce;
          if (!cell_lookup(g, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce))
            continue;
          for (int t = cs; t < ce; t++) {
            const int4 c = g.spts[t];
            const int ex = c.x - q[0], ey = c.y - q[1], ez = c.z - q[2];
            const uint32_t d2 = (uint32_t)(ex * ex) + (uint32_t)(ey * ey) + (uint32_t)(ez * ez);
            uint64_t key = ((uint64_t)d2 << 32) | (uint32_t)c.w;
            if (key < best[KC - 1]) {
              int pay = t;
#pragma unroll
              for (int j = 0; j < KC; j++) {
                const bool lt = key < best[j];
                const uint64_t hi = lt ? best[j] : key;
                const int ph = lt ? bpos[j] : pay;
                best[j] = lt ? key : best[j];
                bpos[j] = lt ? pay : bpos[j];
                key = hi;
                pay = ph;
              }
            }
            runner = key < runner ? key : runner;  // key: the candidate itself if it was not admitted, else what it pushed out
            if ((double)d2 < r2)
              moments_add(m, c.x, c.y, c.z);
          }
        }
      }
    }
    uint64_t R2;
    const bool bounded = guaranteed_radius(g, q, ci, rho, R2);
    if (!bounded) {
      done = true;
    } else {
      uint64_t kth = ~0ull;
#pragma unroll
      for (int j = 0; j < KC; j++)
        kth = (j == K - 1) ? best[j] : kth;
      const bool knn_ok = kth != ~0ull && (kth >> 32) < R2;
      const bool nrm_ok = (double)R2 >= r2;
      done = knn_ok && nrm_ok;
    }
  }
  uint64_t kth_final = ~0ull;
#pragma unroll
  for (int j = 0; j < KC; j++)
    kth_final = (j == K - 1) ? best[j] : kth_final;
  if (!done || m.n > max_nn || kth_final == ~0ull) {
    const int slot = atomicAdd(fb_count, 1);
    fb_list[slot] = (int32_t)s;
    return;
  }
  int32_t* row = neigh + (int64_t)(loc - q_begin) * K;
#pragma unroll
  for (int j = 0; j < KC; j++)
    if (j < K)
      row[j] = (int32_t)(uint32_t)best[j];
  if (npos) {  // rows in POSITION order: coalesced
    int32_t* prow = npos + s * K;
#pragma unroll
    for (int j = 0; j < KC; j++)
      if (j < K)
        prow[j] = bpos[j];
  }
  if (normals) {
    const V3 nv = normal_from_moments(m);
    double* o = normals + 3 * (int64_t)(loc - q_begin);
    o[0] = nv.x;
    o[1] = nv.y;
    o[2] = nv.z;
    if (npos) {  // ... and once more in position order, for the grower's records
      double* po = pnorm_of(npos, g.n, K) + 3 * s;
      po[0] = nv.x;
      po[1] = nv.y;
      po[2] = nv.z;
    }
  }
  if (cert_r2 && (kth_final >> 32) >= cert_r2)
    atomicAdd(uncert, 1ull);
  if (tie_rows) {
    // tie exposure: an equal-d^2 pair inside the k-list or at its boundary -- the only rows where
    // the reference's kd-tree traversal order can differ from this build's canonical (d^2, index) order
    bool tie = false;
    uint64_t next = runner;  // the (K+1)-th neighbour: slot K of the register list when K < KC
#pragma unroll
    for (int j = 0; j + 1 < KC; j++) {
      tie = tie || (j + 1 < K && (best[j] >> 32) == (best[j + 1] >> 32));
      next = (j + 1 == K) ? best[j + 1] : next;
    }
    tie = tie || (next != ~0ull && (next >> 32) == (kth_final >> 32));
    if (tie)
      atomicAdd(tie_rows, 1ull);
  }
}

// (-DBS_KNN_CAP4=1 holds the <16, REL32> instance to 128 registers, 4 waves per SIMD, at the price of 48 bytes of
// scratch: uniform 10 M 10.8 against 11.5 ms, urban 50 M 27.7 against 27.3 ms -- and 27.7 instead of 14.7 GB of
// fabric traffic per 50 M pass (PMC): the spills; not the default)
#ifndef BS_KNN_CAP4
#define BS_KNN_CAP4 0
#endif
constexpr int KBUF = 8;  // waiting candidates per thread of knn_fast2_kernel

// REL32: the moment sums of the d^2 < r^2 ball are kept RELATIVE to the query in 32-bit registers (r <= 181 mm: every
// product is below 2^15, a ball the fast path accepts holds at most max_nn <= 64 points) and turned into the absolute
// 64-bit sums at the end -- exactly: sum x = n q + sum e, sum x^2 = n q^2 + 2 q sum e + sum e^2, sum xy = n qx qy +
// qx sum ey + qy sum ex + sum ex ey.  Ten registers instead of nineteen and 24-bit multiplies instead of 64-bit
// multiply-adds: the kernel drops below 128 registers (4 waves per SIMD instead of 3).
template <int KC, bool REL32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KC == 16 && REL32 && BS_KNN_CAP4 ? 4 : 2))) void knn_fast2_kernel(GridDev g, int64_t q_begin, int64_t q_end, int K,
                                                       int max_nn, double r2, int32_t* __restrict__ neigh,
                                                       double* __restrict__ normals,
                                                       int32_t* __restrict__ fb_list,
                                                       int32_t* __restrict__ fb_count, uint64_t cert_r2,
                                                       unsigned long long* __restrict__ uncert,
                                                       int32_t* __restrict__ npos, unsigned long long* __restrict__ tie_rows)
{
  const int64_t s = xcd_logical_block() * (int64_t)blockDim.x + threadIdx.x;
  if (s >= g.n)
    return;
  const int32_t loc = g.slocal[s];
  if (loc < q_begin || loc >= q_end)
    return;
  // runner-up: the smallest key that is NOT in the list (only needed when the list fills every slot, K == KC):
  // the (K+1)-th neighbour decides whether the list's boundary is an equal-d^2 tie
  uint64_t runner = ~0ull;
  const int4 P = g.spts[s];
  const int q[3] = {P.x, P.y, P.z};
  const int ci[3] = {(int)((uint32_t)(P.x - g.mn[0]) / (uint32_t)g.cell),
                     (int)((uint32_t)(P.y - g.mn[1]) / (uint32_t)g.cell),
                     (int)((uint32_t)(P.z - g.mn[2]) / (uint32_t)g.cell)};
  // candidates that beat the list's last entry wait in LDS (KBUF per thread, slot-major: conflict-free) ...
  __shared__ uint32_t blo[KBUF][256], bhi[KBUF][256];
  __shared__ int bps[KBUF][256];
  int nbuf = 0;
  uint64_t best[KC];
  // cell-sorted POSITION of every kept candidate, carried through the insertion network as a payload:
  // the region grower addresses points by position (bs_grow_spec.hip) and would otherwise have to look
  // every neighbour up by its index -- one random HBM access per edge
  int bpos[KC];
#pragma unroll
  for (int j = 0; j < KC; j++) {
    best[j] = ~0ull;
    bpos[j] = 0;
  }
  Moments m = {};
  int rsx = 0, rsy = 0, rsz = 0, rsxx = 0, rsyy = 0, rszz = 0, rsxy = 0, rsxz = 0, rsyz = 0, rn = 0;
  bool done = false;
  // ... until one thread of the wave has KBUF of them: then every thread sorts its waiting candidates (19
  // compare-exchanges) and merges them into its list (reverse, lower half, bitonic merge: KBUF + KC/2 log2 KC more).
  // The per-candidate insertion network this replaces (KC compare-exchanges, run by the WHOLE wave whenever ONE of its
  // 64 queries admits a candidate -- nearly always) cost 3x as many instructions per query.
  auto flush = [&]() {
    uint64_t bk[KBUF];
    int bp[KBUF];
#pragma unroll
    for (int i = 0; i < KBUF; i++) {
      const bool have = i < nbuf;
      bk[i] = have ? (((uint64_t)bhi[i][threadIdx.x] << 32) | blo[i][threadIdx.x]) : ~0ull;
      bp[i] = have ? bps[i][threadIdx.x] : 0;
    }
    nbuf = 0;
#define BS_CE(ka, pa, kb, pb)            \
  do {                                   \
    const bool sw_ = (kb) < (ka);        \
    const uint64_t tk_ = (ka);           \
    const int tp_ = (pa);                \
    (ka) = sw_ ? (kb) : (ka);            \
    (pa) = sw_ ? (pb) : (pa);            \
    (kb) = sw_ ? tk_ : (kb);             \
    (pb) = sw_ ? tp_ : (pb);             \
  } while (0)
    // optimal 8-input sorting network
    constexpr int net[19][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 2}, {1, 3}, {4, 6}, {5, 7}, {1, 2}, {5, 6},
                                {0, 4}, {3, 7}, {1, 5}, {2, 6}, {1, 4}, {3, 6}, {2, 4}, {3, 5}, {3, 4}};
#pragma unroll
    for (int c = 0; c < 19; c++)
      BS_CE(bk[net[c][0]], bp[net[c][0]], bk[net[c][1]], bp[net[c][1]]);
    // list (ascending) against the waiting candidates (descending): the lower of each pair stays -- a bitonic sequence
    // of the KC smallest; the higher ones are out for good, the smallest of them is the runner-up so far
#pragma unroll
    for (int i = 0; i < KBUF; i++)
      BS_CE(best[KC - KBUF + i], bpos[KC - KBUF + i], bk[KBUF - 1 - i], bp[KBUF - 1 - i]);
#pragma unroll
    for (int i = 0; i < KBUF; i++)
      runner = bk[i] < runner ? bk[i] : runner;
#pragma unroll
    for (int d = KC / 2; d >= 1; d >>= 1) {
#pragma unroll
      for (int i = 0; i < KC; i++)
        if ((i & d) == 0)
          BS_CE(best[i], bpos[i], best[i + d], bpos[i + d]);
    }
#undef BS_CE
  };
  for (int rho = 0; rho <= BS_FAST_RINGS && !done; rho++) {
    for (int dz = -rho; dz <= rho; dz++) {
      const int cz = ci[2] + dz;
      if (cz < 0 || cz >= g.dim[2])
        continue;
      for (int dy = -rho; dy <= rho; dy++) {
        const int cy = ci[1] + dy;
        if (cy < 0 || cy >= g.dim[1])
          continue;
        const bool face = (dz == -rho || dz == rho || dy == -rho || dy == rho);
        const int step = face ? 1 : (rho > 0 ? 2 * rho : 1);
        // The kernel is bound by latency, not by arithmetic (108-128 registers: 4 waves per SIMD, and every hash
        // probe and every candidate used to be ONE load followed by its use): the first probes of up to three cells
        // of a row are issued together, and the candidates of a cell are fetched four at a time.
        for (int dx0 = -rho; dx0 <= rho; dx0 += 3 * step) {
          int4 raw[3];
          uint64_t ck[3];
          uint32_t hh[3];
          bool want[3];
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const int dx = dx0 + u * step;
            const int cx = ci[0] + dx;
            want[u] = dx <= rho && cx >= 0 && cx < g.dim[0];
            ck[u] = pack_cell((uint32_t)(want[u] ? cx : ci[0]), (uint32_t)cy, (uint32_t)cz);
            hh[u] = hash_cell(ck[u]) & g.hmask;
            raw[u] = *reinterpret_cast<const int4*>(&g.table[hh[u]]);
          }
          int cs3[3], ce3[3];
#pragma unroll
          for (int u = 0; u < 3; u++) {
            cs3[u] = 0;
            ce3[u] = 0;
            if (!want[u])
              continue;
            for (;;) {  // (linear probing: the first probe is nearly always the cell or an empty slot)
              const uint64_t ek = (uint64_t)(uint32_t)raw[u].x | ((uint64_t)(uint32_t)raw[u].y << 32);
              if (ek == ck[u]) {
                cs3[u] = raw[u].z;
                ce3[u] = raw[u].w;
                break;
              }
              if (ek == ~0ull)
                break;
              hh[u] = (hh[u] + 1) & g.hmask;
              raw[u] = *reinterpret_cast<const int4*>(&g.table[hh[u]]);
            }
          }
          // ONE copy of the candidate loop (and of the merge in it: 560 instructions) for the three cells
#pragma unroll 1
          for (int u = 0; u < 3; u++) {
            const int cs = u == 0 ? cs3[0] : (u == 1 ? cs3[1] : cs3[2]);
            const int ce = u == 0 ? ce3[0] : (u == 1 ? ce3[1] : ce3[2]);
            for (int t0 = cs; t0 < ce; t0 += 4) {
              int4 cc[4];
#pragma unroll
              for (int v = 0; v < 4; v++)
                cc[v] = g.spts[t0 + v < ce ? t0 + v : ce - 1];
#pragma unroll
              for (int v = 0; v < 4; v++) {
                const int t = t0 + v;
                const int4 c = cc[v];
                const int ex = c.x - q[0], ey = c.y - q[1], ez = c.z - q[2];
                const uint32_t d2 = (uint32_t)(ex * ex) + (uint32_t)(ey * ey) + (uint32_t)(ez * ez);
                const uint64_t key = ((uint64_t)d2 << 32) | (uint32_t)c.w;
                const bool real = t < ce;  // (the last batch of a cell repeats its last point)
                if (real && key < best[KC - 1]) {  // (the list's last entry as of the last merge: never below the true bound)
                  blo[nbuf][threadIdx.x] = (uint32_t)key;
                  bhi[nbuf][threadIdx.x] = d2;
                  bps[nbuf][threadIdx.x] = t;
                  nbuf++;
                } else if (real) {
                  runner = key < runner ? key : runner;  // not admitted
                }
                if (__ballot(nbuf == KBUF))
                  flush();
                if (real && (double)d2 < r2) {
                  if constexpr (REL32) {
                    rsx += ex;
                    rsy += ey;
                    rsz += ez;
                    rsxx += __mul24(ex, ex);
                    rsyy += __mul24(ey, ey);
                    rszz += __mul24(ez, ez);
                    rsxy += __mul24(ex, ey);
                    rsxz += __mul24(ex, ez);
                    rsyz += __mul24(ey, ez);
                    rn += 1;
                  } else {
                    moments_add(m, c.x, c.y, c.z);
                  }
                }
              }
            }
          }
        }
      }
    }
    if (__ballot(nbuf > 0))
      flush();
    uint64_t R2;
    const bool bounded = guaranteed_radius(g, q, ci, rho, R2);
    if (!bounded) {
      done = true;
    } else {
      uint64_t kth = ~0ull;
#pragma unroll
      for (int j = 0; j < KC; j++)
        kth = (j == K - 1) ? best[j] : kth;
      const bool knn_ok = kth != ~0ull && (kth >> 32) < R2;
      const bool nrm_ok = (double)R2 >= r2;
      done = knn_ok && nrm_ok;
    }
  }
  uint64_t kth_final = ~0ull;
#pragma unroll
  for (int j = 0; j < KC; j++)
    kth_final = (j == K - 1) ? best[j] : kth_final;
  if constexpr (REL32) {
    const int64_t n64 = rn, qx = q[0], qy = q[1], qz = q[2];
    m.n = rn;
    m.sx = n64 * qx + rsx;
    m.sy = n64 * qy + rsy;
    m.sz = n64 * qz + rsz;
    m.sxx = (uint64_t)(n64 * qx * qx + 2 * qx * rsx + rsxx);
    m.syy = (uint64_t)(n64 * qy * qy + 2 * qy * rsy + rsyy);
    m.szz = (uint64_t)(n64 * qz * qz + 2 * qz * rsz + rszz);
    m.sxy = n64 * qx * qy + qx * rsy + qy * rsx + rsxy;
    m.sxz = n64 * qx * qz + qx * rsz + qz * rsx + rsxz;
    m.syz = n64 * qy * qz + qy * rsz + qz * rsy + rsyz;
  }
  if (!done || m.n > max_nn || kth_final == ~0ull) {
    const int slot = atomicAdd(fb_count, 1);
    fb_list[slot] = (int32_t)s;
    return;
  }
  int32_t* row = neigh + (int64_t)(loc - q_begin) * K;
#pragma unroll
  for (int j = 0; j < KC; j++)
    if (j < K)
      row[j] = (int32_t)(uint32_t)best[j];
  if (npos) {  // rows in POSITION order: coalesced
    int32_t* prow = npos + s * K;
#pragma unroll
    for (int j = 0; j < KC; j++)
      if (j < K)
        prow[j] = bpos[j];
  }
  if (normals) {
    const V3 nv = normal_from_moments(m);
    double* o = normals + 3 * (int64_t)(loc - q_begin);
    o[0] = nv.x;
    o[1] = nv.y;
    o[2] = nv.z;
    if (npos) {  // ... and once more in position order, for the grower's records
      double* po = pnorm_of(npos, g.n, K) + 3 * s;
      po[0] = nv.x;
      po[1] = nv.y;
      po[2] = nv.z;
    }
  }
  if (cert_r2 && (kth_final >> 32) >= cert_r2)
    atomicAdd(uncert, 1ull);
  if (tie_rows) {
    // tie exposure: an equal-d^2 pair inside the k-list or at its boundary -- the only rows where
    // the reference's kd-tree traversal order can differ from this build's canonical (d^2, index) order
    bool tie = false;
    uint64_t next = runner;  // the (K+1)-th neighbour: slot K of the register list when K < KC
#pragma unroll
    for (int j = 0; j + 1 < KC; j++) {
      tie = tie || (j + 1 < K && (best[j] >> 32) == (best[j + 1] >> 32));
      next = (j + 1 == K) ? best[j + 1] : next;
    }
    tie = tie || (next != ~0ull && (next >> 32) == (kth_final >> 32));
    if (tie)
      atomicAdd(tie_rows, 1ull);
  }
}

// ---- general exact path: explicit sorted lists in scratch -------------------

__device__ inline bool cand_less(uint64_t d2a, int32_t ia, uint64_t d2b, int32_t ib)
{
  return d2a < d2b || (d2a == d2b && ia < ib);
}

// (*out_d2: the smallest d^2 that did NOT stay in the list so far -- the candidate itself or what it pushed out)
__device__ inline void list_insert(uint64_t* d2s, int32_t* idxs, int32_t* poss, int& cnt, int cap, uint64_t d2, int32_t idx,
                                   int32_t pos, uint64_t* out_d2 = nullptr)
{
  int c = cnt;
  if (c == cap) {
    if (!cand_less(d2, idx, d2s[cap - 1], idxs[cap - 1])) {
      if (out_d2 && d2 < *out_d2)
        *out_d2 = d2;
      return;
    }
    if (out_d2 && d2s[cap - 1] < *out_d2)
      *out_d2 = d2s[cap - 1];
    c = cap - 1;
  }
  int j = c;
  while (j > 0 && cand_less(d2, idx, d2s[j - 1], idxs[j - 1])) {
    d2s[j] = d2s[j - 1];
    idxs[j] = idxs[j - 1];
    poss[j] = poss[j - 1];
    j--;
  }
  d2s[j] = d2;
  idxs[j] = idx;
  poss[j] = pos;
  cnt = c + 1;
}

// hybrid (radius) list: sorted by (d^2, global index); pos = cell-sorted position
__device__ inline void hybrid_insert(uint64_t* d2s, int32_t* gids, int32_t* poss, int& cnt, int cap,
                                     uint64_t d2, int32_t gid, int32_t pos)
{
  int c = cnt;
  if (c == cap) {
    if (!cand_less(d2, gid, d2s[cap - 1], gids[cap - 1]))
      return;
    c = cap - 1;
  }
  int j = c;
  while (j > 0 && cand_less(d2, gid, d2s[j - 1], gids[j - 1])) {
    d2s[j] = d2s[j - 1];
    gids[j] = gids[j - 1];
    poss[j] = poss[j - 1];
    j--;
  }
  d2s[j] = d2;
  gids[j] = gid;
  poss[j] = pos;
  cnt = c + 1;
}

__global__ __launch_bounds__(64) void knn_general_kernel(GridDev g, int64_t q_begin, int K, int max_nn,
                                                         double r2, int32_t* __restrict__ neigh,
                                                         double* __restrict__ normals,
                                                         const int32_t* __restrict__ fb_list,
                                                         const int32_t* __restrict__ fb_count,
                                                         uint64_t cert_r2,
                                                         unsigned long long* __restrict__ uncert,
                                                         int32_t* __restrict__ npos, unsigned long long* __restrict__ tie_rows)
{
  const int total = *fb_count;
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < total; w += gridDim.x * blockDim.x) {
    const int32_t s = fb_list[w];
    const int32_t loc = g.slocal[s];
    const int4 P = g.spts[s];
    const int q[3] = {P.x, P.y, P.z};
    const int ci[3] = {(int)((uint32_t)(P.x - g.mn[0]) / (uint32_t)g.cell),
                       (int)((uint32_t)(P.y - g.mn[1]) / (uint32_t)g.cell),
                       (int)((uint32_t)(P.z - g.mn[2]) / (uint32_t)g.cell)};
    uint64_t kd2[32], md2[64];
    int32_t kidx[32], kpos[32], mgid[64], mpos[64];
    int kc = 0, mc = 0;
    uint64_t runner = ~0ull;  // d^2 of the (K+1)-th neighbour once the scan is complete
    bool done = false;
    for (int rho = 0; rho <= BS_GENERAL_RINGS && !done; rho++) {
      for (int dz = -rho; dz <= rho; dz++) {
        const int cz = ci[2] + dz;
        if (cz < 0 || cz >= g.dim[2])
          continue;
        for (int dy = -rho; dy <= rho; dy++) {
          const int cy = ci[1] + dy;
          if (cy < 0 || cy >= g.dim[1])
            continue;
          const bool face = (dz == -rho || dz == rho || dy == -rho || dy == rho);
          const int step = face ? 1 : (rho > 0 ? 2 * rho : 1);
          for (int dx = -rho; dx <= rho; dx += step) {
            const int cx = ci[0] + dx;
            if (cx < 0 || cx >= g.dim[0])
              continue;
            int cs, ce;
            if (!cell_lookup(g, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce))
              continue;
            for (int t = cs; t < ce; t++) {
              const int4 c = g.spts[t];
              const int64_t ex = (int64_t)c.x - q[0], ey = (int64_t)c.y - q[1], ez = (int64_t)c.z - q[2];
              const uint64_t d2 = (uint64_t)(ex * ex) + (uint64_t)(ey * ey) + (uint64_t)(ez * ez);
              list_insert(kd2, kidx, kpos, kc, K, d2, c.w, t, &runner);
              if ((double)d2 < r2)
                hybrid_insert(md2, mgid, mpos, mc, max_nn, d2, c.w, t);
            }
          }
        }
      }
      uint64_t R2;
      const bool bounded = guaranteed_radius(g, q, ci, rho, R2);
      if (!bounded)
        done = true;
      else
        done = (kc == K) && (kd2[K - 1] < R2) && ((double)R2 >= r2);
    }
    if (!done) {
      // full scan of the cloud: always exact
      kc = 0;
      mc = 0;
      runner = ~0ull;
      for (int64_t t = 0; t < g.n; t++) {
        const int4 c = g.spts[t];
        const int64_t ex = (int64_t)c.x - q[0], ey = (int64_t)c.y - q[1], ez = (int64_t)c.z - q[2];
        const uint64_t d2 = (uint64_t)(ex * ex) + (uint64_t)(ey * ey) + (uint64_t)(ez * ez);
        list_insert(kd2, kidx, kpos, kc, K, d2, c.w, (int32_t)t, &runner);
        if ((double)d2 < r2)
          hybrid_insert(md2, mgid, mpos, mc, max_nn, d2, c.w, (int32_t)t);
      }
    }
    int32_t* row = neigh + (int64_t)(loc - q_begin) * K;
    for (int j = 0; j < K; j++)
      row[j] = kidx[j];
    if (npos)
      for (int j = 0; j < K; j++)
        npos[(int64_t)s * K + j] = kpos[j];
    if (normals) {
      Moments m = {};
      for (int j = 0; j < mc; j++) {
        const int4 c = g.spts[mpos[j]];
        moments_add(m, c.x, c.y, c.z);
      }
      const V3 nv = normal_from_moments(m);
      double* o = normals + 3 * (int64_t)(loc - q_begin);
      o[0] = nv.x;
      o[1] = nv.y;
      o[2] = nv.z;
      if (npos) {
        double* po = pnorm_of(npos, g.n, K) + 3 * (int64_t)s;
        po[0] = nv.x;
        po[1] = nv.y;
        po[2] = nv.z;
      }
    }
    if (cert_r2 && kd2[K - 1] >= cert_r2)
      atomicAdd(uncert, 1ull);
    if (tie_rows) {
      bool tie = runner == kd2[K - 1];
      for (int j = 0; j + 1 < K; j++)
        tie = tie || kd2[j] == kd2[j + 1];
      if (tie)
        atomicAdd(tie_rows, 1ull);
    }
  }
}

// ---- general exact path by the WAVE ------------------------------------------------------------------------------
// One wave per work-listed query (a query of the fast kernel that holds more than max_nn points inside the radius --
// the hybrid search keeps the nearest max_nn, which needs an order -- or was not certified within its rings).  The
// lanes share the cells of a ring, every point they see goes into an LDS buffer, the buffer is sorted by (d^2, index)
// with a bitonic network after each ring: the k-list is its first K entries, the hybrid list the entries with
// d^2 < r^2 among its first max_nn, the runner-up entry K.  A query whose candidates do not fit, or that is still
// uncertified after BS_GENERAL_RINGS rings, goes to the one-thread kernel above (second work list).  A thread of
// that kernel took milliseconds per query (two insertion sorts in scratch memory over a few hundred candidates):
// 0.4 % of a uniform cloud's queries were 40 % of its stage-2 time.
constexpr int GCAP = 2048;  // candidates per query in LDS (32 KB)

__global__ __launch_bounds__(64) void knn_general_wave_kernel(GridDev g, int64_t q_begin, int K, int max_nn, double r2,
                                                              int32_t* __restrict__ neigh, double* __restrict__ normals,
                                                              const int32_t* __restrict__ fb_list,
                                                              const int32_t* __restrict__ fb_count, uint64_t cert_r2,
                                                              unsigned long long* __restrict__ uncert,
                                                              int32_t* __restrict__ npos, unsigned long long* __restrict__ tie_rows,
                                                              int32_t* __restrict__ fb2_list, int32_t* __restrict__ fb2_count)
{
  __shared__ uint64_t bd2[GCAP];
  __shared__ int32_t bidx[GCAP], bpos[GCAP];
  __shared__ int cxyz[64][3];
  const int lane = threadIdx.x;
  const int total = *fb_count;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int32_t s = fb_list[w];
    const int32_t loc = g.slocal[s];
    const int4 P = g.spts[s];
    const int q[3] = {P.x, P.y, P.z};
    const int ci[3] = {(int)((uint32_t)(P.x - g.mn[0]) / (uint32_t)g.cell),
                       (int)((uint32_t)(P.y - g.mn[1]) / (uint32_t)g.cell),
                       (int)((uint32_t)(P.z - g.mn[2]) / (uint32_t)g.cell)};
    int count = 0;
    bool done = false, overflow = false;
    for (int rho = 0; rho <= BS_GENERAL_RINGS && !done && !overflow; rho++) {
      const int side = 2 * rho + 1, ncell = side * side * side;
      for (int c0 = 0; c0 < ncell; c0 += 64) {  // (uniform trip count)
        const int c = c0 + lane;
        const int dz = c / (side * side) - rho, dy = (c / side) % side - rho, dx = c % side - rho;
        const int cx = ci[0] + dx, cy = ci[1] + dy, cz = ci[2] + dz;
        const int cheb = max(max(abs(dx), abs(dy)), abs(dz));
        int cs = 0, ce = 0;
        if (c < ncell && cheb == rho && cx >= 0 && cx < g.dim[0] && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2]) {
          if (!cell_lookup(g, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce))
            cs = ce = 0;
        }
        for (int t = cs; __ballot(t < ce) != 0; t++) {
          const bool have = t < ce;
          const unsigned long long m = __ballot(have);
          const int slot = count + __popcll(m & ((1ull << lane) - 1ull));
          if (have && slot < GCAP) {
            const int4 p = g.spts[t];
            const int64_t ex = (int64_t)p.x - q[0], ey = (int64_t)p.y - q[1], ez = (int64_t)p.z - q[2];
            bd2[slot] = (uint64_t)(ex * ex) + (uint64_t)(ey * ey) + (uint64_t)(ez * ez);
            bidx[slot] = p.w;
            bpos[slot] = t;
          }
          count += __popcll(m);
        }
      }
      if (count > GCAP) {
        overflow = true;
        break;
      }
      // bitonic sort of the buffer (padded to a power of two with +inf) by (d^2, index), ascending
      int P2 = 64;
      while (P2 < count)
        P2 <<= 1;
      for (int i = count + lane; i < P2; i += 64) {
        bd2[i] = ~0ull;
        bidx[i] = 0x7fffffff;
        bpos[i] = 0;
      }
      __syncthreads();
      for (int k2 = 2; k2 <= P2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
          for (int i = lane; i < P2; i += 64) {
            const int l = i ^ j;
            if (l > i) {
              const uint64_t da = bd2[i], db = bd2[l];
              const int32_t ia = bidx[i], ib = bidx[l];
              const bool a_gt_b = da > db || (da == db && ia > ib);
              const bool up = (i & k2) == 0;
              if (a_gt_b == up) {
                bd2[i] = db;
                bd2[l] = da;
                bidx[i] = ib;
                bidx[l] = ia;
                const int32_t pa = bpos[i];
                bpos[i] = bpos[l];
                bpos[l] = pa;
              }
            }
          }
          __syncthreads();
        }
      }
      uint64_t R2;
      const bool bounded = guaranteed_radius(g, q, ci, rho, R2);
      if (!bounded)
        done = true;
      else
        done = count >= K && bd2[K - 1] < R2 && (double)R2 >= r2;
      __syncthreads();
    }
    if (overflow || !done || count < K) {  // the one-thread kernel scans on (or the whole cloud)
      if (lane == 0)
        fb2_list[atomicAdd(fb2_count, 1)] = s;
      __syncthreads();
      continue;
    }
    if (lane < K) {
      neigh[(int64_t)(loc - q_begin) * K + lane] = bidx[lane];
      if (npos)
        npos[(int64_t)s * K + lane] = bpos[lane];
    }
    if (normals) {
      // hybrid list: the entries with d^2 < r^2 among the first max_nn (the buffer is sorted by d^2: they are a prefix)
      const bool in = lane < max_nn && lane < count && (double)bd2[lane] < r2;
      const int mc = __popcll(__ballot(in));
      if (in) {
        const int4 p = g.spts[bpos[lane]];
        cxyz[lane][0] = p.x;
        cxyz[lane][1] = p.y;
        cxyz[lane][2] = p.z;
      }
      __syncthreads();
      if (lane == 0) {
        Moments m = {};
        for (int j = 0; j < mc; j++)
          moments_add(m, cxyz[j][0], cxyz[j][1], cxyz[j][2]);
        const V3 nv = normal_from_moments(m);
        double* o = normals + 3 * (int64_t)(loc - q_begin);
        o[0] = nv.x;
        o[1] = nv.y;
        o[2] = nv.z;
        if (npos) {
          double* po = pnorm_of(npos, g.n, K) + 3 * (int64_t)s;
          po[0] = nv.x;
          po[1] = nv.y;
          po[2] = nv.z;
        }
      }
    }
    if (lane == 0) {
      if (cert_r2 && bd2[K - 1] >= cert_r2)
        atomicAdd(uncert, 1ull);
      if (tie_rows) {
        bool tie = count > K && bd2[K] == bd2[K - 1];
        for (int j = 0; j + 1 < K; j++)
          tie = tie || bd2[j] == bd2[j + 1];
        if (tie)
          atomicAdd(tie_rows, 1ull);
      }
    }
    __syncthreads();  // (the buffers are reused by the next query)
  }
}

__global__ void mark_all_kernel(GridDev g, int64_t q_begin, int64_t q_end, int32_t* fb_list,
                                int32_t* fb_count)
{
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= g.n)
    return;
  const int32_t loc = g.slocal[s];
  if (loc < q_begin || loc >= q_end)
    return;
  fb_list[atomicAdd(fb_count, 1)] = (int32_t)s;
}

}  // namespace

int launch_knn_normals(bs_ctx* ctx, const GridDev& g, int64_t q_begin, int64_t q_end, const bs_params& p,
                       int32_t* d_neigh, double* d_normals, double cert_radius, int64_t* n_uncertified, int32_t* d_npos)
{
  hipStream_t st = ctx->stream;
  const int64_t n = g.n;
  BS_HIP(ctx, ctx->fb_list.reserve(sizeof(int32_t) * (2 * n + 32)));
  int32_t* fb_list = ctx->fb_list.as<int32_t>() + 16;
  int32_t* fb_count = ctx->fb_list.as<int32_t>();
  int32_t* fb2_list = fb_list + n + 16;  // queries the wave kernel hands on to the one-thread kernel
  int32_t* fb2_count = ctx->fb_list.as<int32_t>() + 2;
  unsigned long long* uncert = (unsigned long long*)(ctx->fb_list.as<int32_t>() + 4);
  unsigned long long* tie_rows = (unsigned long long*)(ctx->fb_list.as<int32_t>() + 8);
  BS_HIP(ctx, hipMemsetAsync(ctx->fb_list.p, 0, 64, st));
  const double r2 = p.radius * p.radius;
  uint64_t cert_r2 = 0;
  if (cert_radius > 0) {
    double c2 = cert_radius * cert_radius;
    cert_r2 = c2 >= 1.8e19 ? ~0ull : (uint64_t)std::ceil(c2);
    if (cert_r2 == 0)
      cert_r2 = 1;
  }
  const int blocks = (int)((n + 255) / 256);
  const int xblocks = xcd_grid(blocks);
  // fast kernel needs every candidate d^2 < 2^32
  const bool fast_ok = (int64_t)g.cell * (2 * BS_FAST_RINGS + 1) <= 37500;
  if (fast_ok) {
    // (an LDS-staged tile variant -- the 27-cell neighbourhood of 256 consecutive queries staged once per workgroup --
    // was measured slower on MI355X, 1 M points: 1.00 vs 0.58 ms, 50 M: 51.7 vs 24.3 ms, and is retired: the cell-sorted
    // order already makes L1 / L2 serve the candidates, while staging costs LDS atomics, three barriers and occupancy)
    // knn_fast2_kernel (waiting candidates merged eight at a time, probes and candidates fetched in batches) against
    // knn_fast_kernel (insertion network per candidate, one load per use), MI355X: urban 10 M 9.7 vs 13.6 ms,
    // uniform 10 M 16.0 vs 18.1 ms, urban 50 M 27.4-28.1 vs 28.8-29.4 ms.  Counters at 50 M (rocprofv3 --pmc): 11.9 k
    // instead of 16.0 k vector instructions per wave, but 135 instead of 108 registers (3 instead of 4 waves per
    // SIMD) -- capped to 128 registers the spills (140 bytes of scratch) cost more than the fourth wave brings
    // (66 ms).  BS_KNN_BUFFERED=0 selects the first kernel.
    const bool buffered = getenv("BS_KNN_BUFFERED") ? atoi(getenv("BS_KNN_BUFFERED")) != 0 : true;
    const bool rel32 = r2 <= 32761.0 && !getenv("BS_KNN_MOMENTS64");  // r <= 181 mm (see REL32)
    if (buffered && p.k <= 16 && rel32)
      knn_fast2_kernel<16, true><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                         fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
    else if (buffered && p.k <= 16)
      knn_fast2_kernel<16, false><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                          fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
    else if (buffered && rel32)
      knn_fast2_kernel<32, true><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                         fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
    else if (buffered)
      knn_fast2_kernel<32, false><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                          fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
    else if (p.k <= 16)
      knn_fast_kernel<16><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                  fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
    else
      knn_fast_kernel<32><<<xblocks, 256, 0, st>>>(g, q_begin, q_end, p.k, p.max_nn, r2, d_neigh, d_normals,
                                                  fb_list, fb_count, cert_r2, uncert, d_npos, tie_rows);
  } else {
    mark_all_kernel<<<blocks, 256, 0, st>>>(g, q_begin, q_end, fb_list, fb_count);
  }
  if (getenv("BS_KNN_GENERAL_THREAD")) {  // developer A/B switch: the one-thread kernel for the whole work list
    knn_general_kernel<<<1024, 64, 0, st>>>(g, q_begin, p.k, p.max_nn, r2, d_neigh, d_normals, fb_list, fb_count, cert_r2,
                                            uncert, d_npos, tie_rows);
  } else {
    knn_general_wave_kernel<<<8192, 64, 0, st>>>(g, q_begin, p.k, p.max_nn, r2, d_neigh, d_normals, fb_list, fb_count, cert_r2,
                                                 uncert, d_npos, tie_rows, fb2_list, fb2_count);
    knn_general_kernel<<<1024, 64, 0, st>>>(g, q_begin, p.k, p.max_nn, r2, d_neigh, d_normals, fb2_list, fb2_count, cert_r2,
                                            uncert, d_npos, tie_rows);
  }
  BS_HIP(ctx, hipGetLastError());
  // bookkeeping read-back (also the point where kernel faults surface)
  int32_t hb[4];
  BS_HIP(ctx, hipMemcpyAsync(hb, ctx->fb_list.p, sizeof hb, hipMemcpyDeviceToHost, st));
  unsigned long long hu = 0, ht = 0;
  BS_HIP(ctx, hipMemcpyAsync(&hu, uncert, sizeof hu, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipMemcpyAsync(&ht, tie_rows, sizeof ht, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  ctx->tm.n_fallback_queries = hb[0];
  ctx->tm.tie_rows = (int64_t)ht;
  if (n_uncertified)
    *n_uncertified = (int64_t)hu;
  return BS_OK;
}

}  // namespace bs
