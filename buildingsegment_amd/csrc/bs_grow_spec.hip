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
//      a triangular system (only lower indices matter) with a unique fixed
//      point.  It is kept up to date dynamically: reverse lists R(c) of the
//      static masks are built once and pull_pass_kernel re-evaluates only the
//      points dirtied by an inserted / dropped plane until nothing flips.
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
//      validates, so every round makes progress.  Consistent planes above that
//      point stay "pending" (kept in the owner structure, re-validated every
//      round, never grown again unless invalidated); short planes that lose a
//      point re-grow inside the same launch.
//
// Final labels: plane_idx[p] = 1 + #(committed planes with seed < owner[p])
// (cur_planeId only advances on commit, :199-202), -1 if owner[p] is none.
//
// Index spaces.  Everything in this file ADDRESSES points by their position in the search grid's
// cell-sorted (Morton) order -- records, masks, owners, reverse lists, neighbour rows, plane lists
// during growth -- so that the neighbours of a point live in nearby memory and the graph passes
// (static masks, reverse lists, owner fixed point, candidate scan) are served by L2 instead of
// one HBM access per edge.  The sequential semantics of the reference are about the ORIGINAL
// indices (seed scan order :184, "earlier attempt wins"): prio[s] = original index of position s is
// the PRIORITY, and every value that is compared -- seeds, owners, claim tags, F -- is an original
// index.  Positions never take part in a comparison; original indices never address anything but
// dead[] (per-seed flags) and the caller's output arrays.  Without a cached grid order (foreign
// input handed to bs_region_grow*) the order is the identity.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <chrono>
#include <vector>

#include "bs_centerdiv.h"
#include "bs_common.h"

namespace bs {

namespace {

constexpr int32_t INF = 0x7fffffff;
constexpr int32_t LOWER_BIT = (int32_t)0x80000000;  // in a reverse-list entry: the source's original index is below the target's
constexpr int32_t POS_MASK = 0x7fffffff;
constexpr int MAX_WAVES = 1 << 18;  // upper bound of plane attempts grown per round (the 50 M cloud has 158 k candidates in round 1)
constexpr int MAX_PENDING = 32768;  // finished planes waiting for earlier attempts

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
  int32_t pad;
};

struct PlaneOut {
  double normal[3];
  int64_t list_off;  // into the round pool
  int64_t list_n;
  int64_t log_off;
  int64_t log_n;
  int64_t steps;
  int32_t center[3];
  int32_t seed;        // original index of the seed
  int32_t status;
  int32_t keep;        // committed (list_n > th_count)
  int32_t consistent;  // set by validate2_kernel
  int32_t pad;         // host command for plane_apply_kernel
  int32_t thief;       // diagnostics: seed of the plane that took a point from this one (-1: none)
  int32_t seed_pos;    // position of the seed (seed itself is the ORIGINAL index: the priority)
  int32_t v3ok;        // validate3 (runs beside the owner passes on a second stream): state reproducible from the list
  int32_t pad4;
  int64_t t_start, t_end;  // wall_clock64() (100 MHz) at the wave's start and end: diagnostics (BS_DEBUG)
  int32_t w;               // number of the attempt in its round (rank by seed)
  int32_t pad5;
};

// what the host's merge looks at after a round: everything but the attempts that failed at depth 0 or gave up
struct KeepForHost {
  __device__ bool operator()(const PlaneOut& o) const { return o.status == ST_DONE || o.status == ST_NOMEM || o.status == ST_WATCHDOG; }
};

struct Pool {
  int32_t* base;
  unsigned long long* top;
  unsigned long long cap;
};

struct Slab {
  int64_t off;
  int32_t cap;  // entries (int32); a slab never exceeds 2^31 - 4 of them
};

__device__ inline int ld_i32(const int32_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-wide vote straight from a condition (HIP's __ballot(int) first materialises the
// predicate as 0/1 in a VGPR and compares it again: two VALU ops per vote)
__device__ inline unsigned long long ballot64(bool pred)
{
  return __builtin_amdgcn_ballot_w64(pred);
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
__device__ inline bool slab_ensure(const Pool& pool, Slab& s, int32_t used, int64_t need, int lane)
{
  if (need <= s.cap)
    return true;
  if (need > 0x7ffffff0ll)
    return false;
  int64_t ncap = (int64_t)s.cap * 2;
  if (ncap < need)
    ncap = need;
  if (ncap < 256)
    ncap = 256;
  if (ncap > 0x7ffffff0ll)
    ncap = 0x7ffffff0ll;
  ncap = (ncap + 3) & ~(int64_t)3;  // keep slab offsets 16-byte aligned (int4 stack slots)
  unsigned long long off = 0;
  if (lane == 0)
    off = atomicAdd(pool.top, (unsigned long long)ncap);
  off = ((unsigned long long)(uint32_t)readlane_i32((int)(off >> 32), 0) << 32) |
        (uint32_t)readlane_i32((int)(off & 0xffffffffu), 0);
  if (off + (unsigned long long)ncap > pool.cap)
    return false;
  for (int32_t t0 = 0; t0 < used; t0 += 64) {  // (uniform trip count: a per-lane loop exit makes the enclosing loops divergent)
    const int32_t t = t0 + lane;
    if (t < used)
      pool.base[off + t] = ld_i32(pool.base + s.off + t);
  }
  s.off = (int64_t)off;
  s.cap = (int32_t)ncap;
  return true;
}

// ---- (a) static depth-0 mask ------------------------------------------------
// order (nullable): a spatially coherent permutation (the grid's cell-sorted
// order), so that the neighbour gathers of adjacent threads share cache lines.
// Neighbour geometry comes from the one-line-per-point records.
// Reverse-list counting / filling without one global atomic per edge.  Integer atomics are executed at
// the memory side (64 uncached bytes each, whatever the locality): the 750 M edges of the 50 M cloud were
// 48 GB of atomic traffic per pass.  In position space a point's neighbours sit within a few hundred
// positions, so every workgroup (256 consecutive positions) counts the edges into its WINDOW of
// RW_WIN positions in LDS and touches global memory once per distinct target; only the rare edge that
// leaves the window takes the global atomic.
constexpr int RW_PAD = 896;
constexpr int RW_WIN = 256 + 2 * RW_PAD;  // 2048 positions
constexpr int GW_PAD = 192;               // geometry tile of static_mask_kernel: 640 positions x 48 B = 30 KB of LDS (38 KB with the counters).  Measured at 50 M: padding 384 (48 KB, two workgroups per CU) 7.97 ms, 192 5.42 ms, 128 5.43 ms
constexpr int GW_WIN = 256 + 2 * GW_PAD;

// Static depth-0 masks.  One workgroup = 256 consecutive positions.  The geometry (position + normal, 48 B)
// of the workgroup's WINDOW of positions is staged once into LDS with coalesced loads; a neighbour inside the
// window (nearly all of them: neighbours in space are neighbours in Morton position) costs an LDS read, the
// rare one outside a global gather.  Before, every one of the k-1 neighbour tests fetched three 16-byte pieces
// of a 128-byte record through L2 -- 1.6 KB of fabric traffic per point.
// (rows / row_stride: the neighbour rows in position space -- the kNN kernels' compact copy (stride K: 64 bytes per
// point) in the fused pipeline, the second half of the 128-byte records otherwise)
__global__ __launch_bounds__(256) void static_mask_kernel(SpecArgs a, const int32_t* __restrict__ rows, int row_stride,
                                                          const int4* __restrict__ geo, uint32_t* __restrict__ hmask,
                                                          int32_t* __restrict__ rcnt, uint32_t* __restrict__ lmask)
{
  __shared__ int lcnt[RW_WIN];
  __shared__ int4 lgeo[GW_WIN * 3];
  const int64_t b0 = xcd_logical_block() * (int64_t)blockDim.x;
  const int64_t i = b0 + threadIdx.x;  // position: spatial neighbours are adjacent threads
  const int64_t w0 = b0 > RW_PAD ? b0 - RW_PAD : 0;
  const int64_t g0 = b0 > GW_PAD ? b0 - GW_PAD : 0;
  for (int d = threadIdx.x; d < RW_WIN; d += 256)
    lcnt[d] = 0;
  {
    const int64_t gend = (g0 + GW_WIN < a.n ? g0 + GW_WIN : a.n) - g0;  // positions available in the tile
    const int4* src = geo + 3 * g0;
    for (int64_t t = threadIdx.x; t < 3 * gend; t += 256)
      lgeo[t] = src[t];
  }
  __syncthreads();
  if (i < a.n) {
    const int4* own = lgeo + 3 * (i - g0);
    const int4 s0 = own[0], s1 = own[1], s2 = own[2];
    const double cnx = __hiloint2double(s1.y, s1.x), cny = __hiloint2double(s1.w, s1.z),
                 cnz = __hiloint2double(s2.y, s2.x);
    const int ccx = s0.x, ccy = s0.y, ccz = s0.z;
    const int32_t* row = rows + i * row_stride;
    uint32_t m = 0, lm = 0;  // lm: bit t-1 set when this point precedes neighbour t in the original order
    for (int t = 1; t < a.K; t++) {
      const int32_t c = row[t];
      if (c < 0)
        continue;  // a repeated index of a caller-supplied row (see build_records_kernel): no neighbour
      const int64_t gd = (int64_t)c - g0;
      int4 q0, q1, q2;
      if (gd >= 0 && gd < GW_WIN) {
        q0 = lgeo[3 * gd];
        q1 = lgeo[3 * gd + 1];
        q2 = lgeo[3 * gd + 2];
      } else {
        const int4* rc = geo + 3 * (int64_t)c;
        q0 = rc[0];
        q1 = rc[1];
        q2 = rc[2];
      }
      const int dx = (int)((uint32_t)q0.x - (uint32_t)ccx);
      const int dy = (int)((uint32_t)q0.y - (uint32_t)ccy);
      const int dz = (int)((uint32_t)q0.z - (uint32_t)ccz);
      const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
      const double dt = cnx * __hiloint2double(q1.y, q1.x) + cny * __hiloint2double(q1.w, q1.z) +
                        cnz * __hiloint2double(q2.y, q2.x);
      if (dist <= a.th && dt >= a.cos_th) {
        m |= 1u << (t - 1);
        lm |= (s0.w < q0.w ? 1u : 0u) << (t - 1);
        const int64_t d = (int64_t)c - w0;  // |R(c)|: the reverse lists are counted in the same pass
        if (d >= 0 && d < RW_WIN)
          atomicAdd(&lcnt[d], 1);
        else
          atomicAdd(&rcnt[c], 1);
      }
    }
    hmask[i] = m;
    lmask[i] = lm;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RW_WIN; d += 256) {
    const int v = lcnt[d];
    if (v)
      atomicAdd(&rcnt[w0 + d], v);
  }
}

// ---- orphan-maker fixed point (dynamic, pull based) ----------------------------
// owner[c] = min( base[c], min{ j in R(c) : occ[j] } ),   occ[j] = maker(j) && owner[j] >= j
// R(c) = reverse lists of the static masks (who may claim c), built once.  The
// state (owner, occ) persists over the whole call; inserting or dropping a plane
// only marks the touched points dirty and pull_pass_kernel re-evaluates dirty
// points until nothing flips.  Dependencies only run from lower to higher
// indices, so the iteration settles bottom-up to the unique fixed point; work is
// proportional to what actually changes, not to n.
__global__ __launch_bounds__(256) void rev_fill_kernel(const uint32_t* __restrict__ hmask, const int32_t* __restrict__ rows,
                                                       int row_stride, int64_t n, unsigned long long* __restrict__ rcur,
                                                       int32_t* __restrict__ radj, const uint32_t* __restrict__ lmask)
{
  // same window as static_mask_kernel: count per target in LDS, reserve ONE range per (workgroup, target)
  // with a global atomic on the target's fill cursor (rcur[c] starts at roff[c]), then hand out the slots
  // of the range through LDS.  The order of a reverse list is irrelevant (only its minimum is ever taken).
  __shared__ int lcnt[RW_WIN];
  __shared__ unsigned long long lbase[RW_WIN];
  const int64_t b0 = xcd_logical_block() * (int64_t)blockDim.x;
  const int64_t i = b0 + threadIdx.x;
  const int64_t w0 = b0 > RW_PAD ? b0 - RW_PAD : 0;
  for (int d = threadIdx.x; d < RW_WIN; d += 256)
    lcnt[d] = 0;
  __syncthreads();
  const uint32_t m0 = i < n ? hmask[i] : 0u;
  const uint32_t lm0 = i < n ? lmask[i] : 0u;  // an entry of R(c) carries LOWER_BIT when its source precedes c in the original order
  const int32_t* row = rows + (i < n ? i : 0) * row_stride;
  for (uint32_t m = m0; m;) {
    const int t = __ffs(m) - 1;
    m &= m - 1;
    const int64_t d = (int64_t)row[t + 1] - w0;
    if (d >= 0 && d < RW_WIN)
      atomicAdd(&lcnt[d], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RW_WIN; d += 256) {
    const int v = lcnt[d];
    if (v) {
      lbase[d] = atomicAdd(&rcur[w0 + d], (unsigned long long)v);
      lcnt[d] = 0;
    }
  }
  __syncthreads();
  for (uint32_t m = m0; m;) {
    const int t = __ffs(m) - 1;
    m &= m - 1;
    const int32_t c = row[t + 1];
    const int64_t d = (int64_t)c - w0;
    if (d >= 0 && d < RW_WIN)
      radj[lbase[d] + (unsigned)atomicAdd(&lcnt[d], 1)] = (int32_t)i | (((lm0 >> t) & 1u) ? LOWER_BIT : 0);
    else
      radj[atomicAdd(&rcur[c], 1ull)] = (int32_t)i | (((lm0 >> t) & 1u) ? LOWER_BIT : 0);
  }
}

__global__ void pull_pass_kernel(int64_t n, int K, int sub, const uint32_t* __restrict__ hmask,
                                 const int32_t* __restrict__ prio, const uint8_t* __restrict__ ps,
                                 const int32_t* __restrict__ base, const int64_t* __restrict__ roff,
                                 const int32_t* __restrict__ radj, int32_t* __restrict__ omega, uint32_t* occ,
                                 uint8_t* dirty_cur, uint8_t* dirty_next, uint8_t* bdirty_cur, uint8_t* bdirty_next,
                                 int4* rec, int quads, int* any, const int32_t* __restrict__ minr, const int* prev_any)
{
  // Only a pass that flipped something leaves dirty points behind: if the previous pass of this group reports none,
  // there is nothing to do -- every workgroup leaves after ONE scalar load (the passes behind the settling one of a
  // group of 16 cost 14 us each at 50 M points just for looking at their dirty flags)
  if (prev_any && *prev_any == 0)
    return;
  // Dirty flags on two levels: one per point and one per 256 points.  A workgroup owns `sub`
  // (<= 64) consecutive 256-point groups and visits only the dirty ones: late passes touch a few
  // points of a 50 M cloud, and one workgroup per 256 points cost 0.19 ms per pass in block
  // launches alone (260 passes per segmentation).
  __shared__ unsigned long long sub_mask;
  const int64_t nb256 = (n + 255) >> 8;
  const int64_t g0 = (int64_t)blockIdx.x * sub;
  if (threadIdx.x < 64) {
    const int64_t gidx = g0 + threadIdx.x;
    const bool d = (int)threadIdx.x < sub && gidx < nb256 && bdirty_cur[gidx] != 0;
    if (d)
      bdirty_cur[gidx] = 0;
    const unsigned long long m = ballot64(d);
    if (threadIdx.x == 0)
      sub_mask = m;
  }
  __syncthreads();
  unsigned long long gm = sub_mask;
  bool flipped = false;
  while (gm) {
    const int t = __ffsll(gm) - 1;
    gm &= gm - 1;
    const int64_t c = ((g0 + t) << 8) + threadIdx.x;
    if (c >= n || !dirty_cur[c])
      continue;
    dirty_cur[c] = 0;
    int32_t v = base[c];
    // minr[c] = lowest original index in R(c): a base owner at or below it cannot be undercut by any maker,
    // occurring or not -- the points of an inserted plane (base = its seed) skip the reverse list altogether
    if (v > minr[c]) {
      const int64_t e1 = roff[c + 1];
      // four reverse edges at a time, each level of the dependent chain (edge -> occurrence bit -> original
      // index) issued for all four before the first is waited for: a late pass re-evaluates a few hundred
      // points and is bound by this chain's latency (~14 edges x 3 loads one after the other: 30 us per pass)
      for (int64_t e = roff[c]; e < e1; e += 4) {
        int32_t j[4];
        uint32_t ow[4];
        int32_t pj[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
          j[i] = e + i < e1 ? (radj[e + i] & POS_MASK) : -1;
        // occ is a BITMAP (n / 8 bytes: 6 MB at 50 M points, resident in L2 / Infinity Cache), so the ~14
        // random look-ups per re-evaluated point do not go to HBM
#pragma unroll
        for (int i = 0; i < 4; i++)
          ow[i] = j[i] >= 0 ? __hip_atomic_load(occ + (j[i] >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++)
          pj[i] = ((ow[i] >> (j[i] & 31)) & 1u) ? prio[j[i]] : INF;  // makers are compared by ORIGINAL index
#pragma unroll
        for (int i = 0; i < 4; i++)
          v = pj[i] < v ? pj[i] : v;
      }
    }
    // (the makers' original indices stored beside the reverse-list entries, instead of prio[j] for the occurring ones:
    // measured at 50 M -- owner passes 25.9 vs 25.6 ms per pass, rev_fill 8.7 vs 5.7 ms, decide_finish 3.1 vs 2.7 ms:
    // few sources of a list occur, so the look-up it saves is rare and the extra stream is not)
    omega[c] = v;
    reinterpret_cast<int32_t*>(rec + c * quads)[3] = v;  // the growth kernel reads the owner from the record
    const uint32_t m0 = hmask[c];
    const uint8_t want = (m0 != 0 && !(ps[c] & 1) && v >= prio[c]) ? 1 : 0;
    const uint32_t cbit = 1u << (c & 31);
    const uint8_t have = (__hip_atomic_load(occ + (c >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & cbit) ? 1 : 0;
    if (want != have) {
      if (want)
        __hip_atomic_fetch_or(occ + (c >> 5), cbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        __hip_atomic_fetch_and(occ + (c >> 5), ~cbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t* row = reinterpret_cast<const int32_t*>(rec + c * quads + 4);
      uint32_t m = m0;
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        const int32_t q = row[b + 1];
        dirty_next[q] = 1;
        bdirty_next[q >> 8] = 1;
      }
      flipped = true;
    }
  }
  if (flipped)
    *any = 1;
}

// ---- the FIRST fixed point (no plane yet) by decided states ------------------------------
// Before any plane exists, "maker i occurs" only depends on the makers in R(i) with a LOWER
// original index: i occurs iff none of them occurs.  Instead of starting optimistically and
// undoing (the dirty-flag iteration above needed 4 evaluations and 1.6 flips per point on the
// 50 M cloud, every flip marking ~14 neighbours), each point is decided ONCE, as soon as its
// lower in-neighbours are: st = 0 undecided, 1 occurs, 2 does not.  A decision only reads
// decided states, so it is final whatever the timing; states written during a pass may be picked
// up in the same pass.  Afterwards one pass derives every owner from the occurring makers.
__global__ void decide_init_kernel(const uint32_t* __restrict__ hmask, int64_t n, uint8_t* __restrict__ st,
                                   uint8_t* __restrict__ bund)
{
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const bool maker = c < n && hmask[c] != 0;
  if (c < n)
    st[c] = maker ? 0 : 2;
  const int anym = __syncthreads_or(maker);
  if (threadIdx.x == 0)
    bund[blockIdx.x] = anym ? 1 : 0;
}

__global__ void decide_pass_kernel(int64_t n, int sub, const int32_t* __restrict__ prio,
                                   const int64_t* __restrict__ roff, const int32_t* __restrict__ radj, uint8_t* st,
                                   uint8_t* bund, int* any)
{
  __shared__ unsigned long long sub_mask;
  __shared__ int left[64];
  const int64_t nb256 = (n + 255) >> 8;
  const int64_t g0 = (int64_t)blockIdx.x * sub;
  if (threadIdx.x < 64) {
    const int64_t gidx = g0 + threadIdx.x;
    const bool d = (int)threadIdx.x < sub && gidx < nb256 && bund[gidx] != 0;
    left[threadIdx.x] = 0;
    const unsigned long long m = ballot64(d);
    if (threadIdx.x == 0)
      sub_mask = m;
  }
  __syncthreads();
  unsigned long long gm = sub_mask;
  bool undecided_left = false;
  while (gm) {
    const int t = __ffsll(gm) - 1;
    gm &= gm - 1;
    const int64_t c = ((g0 + t) << 8) + threadIdx.x;
    bool mine_left = false;
    if (c < n && __hip_atomic_load(st + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      bool lower_occ = false, lower_und = false;
      const int64_t e1 = roff[c + 1];
      // only the sources that precede c matter (LOWER_BIT, decided when the lists were built: no look-up of
      // their original index); four edges per trip, states loaded for all four before the first is used
      for (int64_t e = roff[c]; e < e1; e += 4) {
        int32_t rj[4];
        uint8_t sj[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
          rj[q] = e + q < e1 ? radj[e + q] : 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
          sj[q] = rj[q] < 0 ? __hip_atomic_load(st + (rj[q] & POS_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)2;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          lower_occ = lower_occ || sj[q] == 1;
          lower_und = lower_und || sj[q] == 0;
        }
      }
      if (lower_occ)
        __hip_atomic_store(st + c, (uint8_t)2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (!lower_und)
        __hip_atomic_store(st + c, (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        mine_left = true;
    }
    if (mine_left)
      left[t] = 1;  // benign race: everybody writes 1
    undecided_left = undecided_left || mine_left;
  }
  __syncthreads();
  if (threadIdx.x < 64 && (int)threadIdx.x < sub && g0 + threadIdx.x < nb256)
    bund[g0 + threadIdx.x] = left[threadIdx.x] ? 1 : 0;
  if (undecided_left)
    *any = 1;
}

// owners and occupancy bits from the decided states (one thread per position, 64 positions per wave)
__global__ void decide_finish_kernel(int64_t n, const int32_t* __restrict__ prio, const int64_t* __restrict__ roff,
                                     const int32_t* __restrict__ radj, const uint8_t* __restrict__ st,
                                     int32_t* __restrict__ omega, uint32_t* __restrict__ occ, int4* rec, int quads,
                                     int32_t* __restrict__ minr)
{
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool oc = false;
  if (c < n) {
    int32_t v = INF, vall = INF;
    const int64_t e1 = roff[c + 1];
    for (int64_t e = roff[c]; e < e1; e++) {
      const int32_t j = radj[e] & POS_MASK;
      const int32_t pj = prio[j];
      vall = pj < vall ? pj : vall;
      if (st[j] == 1)
        v = pj < v ? pj : v;
    }
    minr[c] = vall;
    omega[c] = v;
    reinterpret_cast<int32_t*>(rec + c * quads)[3] = v;
    oc = st[c] == 1;
  }
  const unsigned long long m = ballot64(oc);
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (c - lane) >> 5;  // blockDim is a multiple of 64: the wave covers 64 aligned positions
  if (lane == 0 && (c - lane) < n)
    occ[w0] = (uint32_t)m;
  if (lane == 32 && (c - lane + 32) < n)
    occ[w0 + 1] = (uint32_t)(m >> 32);
}

// BS_VERIFY=1: is (omega, occ) a fixed point of the owner equations?
__global__ void verify_fixpoint_kernel(int64_t n, const uint32_t* __restrict__ hmask, const int32_t* __restrict__ prio,
                                       const uint8_t* __restrict__ ps,
                                       const int32_t* __restrict__ base, const int64_t* __restrict__ roff,
                                       const int32_t* __restrict__ radj, const int32_t* __restrict__ omega,
                                       const uint32_t* __restrict__ occ, int* nbad)
{
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= n)
    return;
  int32_t v = base[c];
  for (int64_t e = roff[c]; e < roff[c + 1]; e++) {
    const int32_t j = radj[e] & POS_MASK;
    const bool oj = hmask[j] != 0 && !(ps[j] & 1) && omega[j] >= prio[j];
    if (oj && prio[j] < v)
      v = prio[j];
  }
  const bool oc = hmask[c] != 0 && !(ps[c] & 1) && omega[c] >= prio[c];
  if (v != omega[c] || (oc ? 1u : 0u) != ((occ[c >> 5] >> (c & 31)) & 1u))
    atomicAdd(nbad, 1);
}

// ---- plane-attempt candidates -------------------------------------------------
// ps[i]: bit 0 = seed of an inserted (pending or committed) plane, bit 1 = can never seed (static depth-0 mask
// incomplete; set once by seed_flags_kernel).  The scan reads this ONE byte per point and the owner / original
// index only of the points that can seed.
__global__ void seed_flags_kernel(const uint32_t* __restrict__ hmask, int64_t n, int K, uint8_t* __restrict__ ps)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const uint32_t full = (K - 1 >= 32) ? 0xffffffffu : ((1u << (K - 1)) - 1u);
  if (i < n)
    ps[i] = hmask[i] == full ? 0 : 2;
}

__global__ void cand_flag_kernel(const int4* __restrict__ rec, int quads, int K,
                                 int64_t n, int32_t F, const int32_t* __restrict__ prio, const uint8_t* __restrict__ ps,
                                 int all_seeds, const int32_t* __restrict__ omega, int32_t* min_idx,
                                 unsigned long long* __restrict__ cand_out, int32_t* cand_count, uint8_t* __restrict__ bcand,
                                 int sub)
{
  // A 256-position group whose points can never seed again -- incomplete static masks, points
  // owned by a FINAL attempt (owner < F; F only grows) -- is skipped for the rest of the call: in the late
  // rounds of a large cloud almost every group is, and the scan costs microseconds instead of a pass over n.
  // A workgroup owns `sub` (<= 64) consecutive groups and visits the live ones (one workgroup per group was
  // 195 k block launches = 0.14 ms per scan at 50 M, twice per round).
  __shared__ unsigned long long sub_mask;
  const int64_t nb256 = (n + 255) >> 8;
  const int64_t g0 = (int64_t)blockIdx.x * sub;
  if (threadIdx.x < 64) {
    const int64_t gidx = g0 + threadIdx.x;
    const bool live = (int)threadIdx.x < sub && gidx < nb256 && bcand[gidx] != 0;
    const unsigned long long m = ballot64(live);
    if (threadIdx.x == 0)
      sub_mask = m;
  }
  __syncthreads();
  unsigned long long gm = sub_mask;
  while (gm) {  // (block-uniform trip count: the barrier below is reached by every thread)
    const int t = __ffsll(gm) - 1;
    gm &= gm - 1;
    const int64_t i = ((g0 + t) << 8) + threadIdx.x;
    bool c = false, alive = false;
    const uint8_t f = i < n ? ps[i] : (uint8_t)2;
    int32_t pi = 0;
    if (!(f & 2)) {
      pi = prio[i];
      const int32_t oi = omega[i];
      alive = !(oi < F);  // (a pending plane's seed counts as alive: the plane can still be dropped)
      if (pi >= F && oi >= pi && (all_seeds || !(f & 1))) {  // (all_seeds: the audit lists committed seeds too)
        // all K-1 neighbours free at the seed's time; four owners per trip, loaded unconditionally (a
        // short-circuit `c && ...` chains up to K-1 dependent loads one after the other)
        c = true;
        const int32_t* row = reinterpret_cast<const int32_t*>(rec + i * quads + 4);
        for (int u = 1; u < K; u += 4) {
          int32_t ov[4];
#pragma unroll
          for (int q = 0; q < 4; q++)
            ov[q] = u + q < K ? omega[row[u + q]] : INF;
#pragma unroll
          for (int q = 0; q < 4; q++)
            c = c && ov[q] >= pi;
        }
      }
    }
    const int any_alive = __syncthreads_or(alive);
    if (threadIdx.x == 0 && !any_alive)
      bcand[g0 + t] = 0;
    if (c && min_idx)
      atomicMin(min_idx, pi);
    if (c && cand_out)  // sparse: appended unordered as (original index << 32 | position), sorted afterwards
      cand_out[atomicAdd(cand_count, 1)] = ((unsigned long long)(uint32_t)pi << 32) | (uint32_t)i;
  }
}

// ---- (b) speculative plane growth: one wavefront per candidate seed -------------
// Data layout for a latency-bound gather: one record per point holding
// everything a Broad() test needs plus the point's own neighbour row, so a
// candidate costs ONE cache line:
//   int4 q0 = x, y, z, owner (tentative owner of this round)
//   int4 q1 = normal.x, normal.y            (f64 bit patterns)
//   int4 q2 = normal.z, tag, pad            (tag: in-flight claim, atomics)
//   int4 q3 = pad
//   int4 q4.. = neighbour row (K ints; slot 0 unused by Broad, :224)
// 128 B per point at K <= 16, 192 B at K <= 32.
template <int KC>
struct RecLayout {
  static constexpr int QUADS = 4 + KC / 4;  // int4 per record
};

__device__ inline int32_t* rec_tag(int4* rec, int quads, int64_t i)
{
  return reinterpret_cast<int32_t*>(rec + i * quads + 2) + 2;
}

__global__ void invert_order_kernel(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ pos)
{
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s < n)
    pos[order[s]] = (int32_t)s;
}

// One record per POSITION s of the cell-sorted order (order[s] = original index, pos = its inverse;
// both null: identity).  The neighbour row is translated to positions here -- the one pass of the
// stage that has to look things up by original index.
template <int KC>
__global__ __launch_bounds__(256) void build_records_kernel(SpecArgs a, const int32_t* __restrict__ order, const int32_t* __restrict__ pos,
                                     const int32_t* __restrict__ npos, int4* __restrict__ rec, int32_t* __restrict__ prio,
                                     int4* __restrict__ geo, const int4* __restrict__ spts, const double* __restrict__ pnorm)
{
  constexpr int Q = RecLayout<KC>::QUADS;
  // Records (128 B) and geometry (48 B) are put together in LDS and leave in whole cache lines: a thread storing its own
  // record issues eight 16-byte stores 128 bytes apart from its neighbours' (2.4 TB/s of mostly partial-line writes)
  __shared__ int4 srec[256 * Q];
  __shared__ int4 sgeo[256 * 3];
  const int64_t b0 = blockIdx.x * (int64_t)blockDim.x;
  const int64_t s = b0 + threadIdx.x;
  if (s < a.n) {
  // fused pipeline: coordinates + original index from the grid's cell-sorted copy, normals from the kNN kernels'
  // position-ordered copy -- everything streams; otherwise both are gathered by original index
  int64_t i;
  int px, py, pz;
  if (spts) {
    const int4 p = spts[s];
    px = p.x;
    py = p.y;
    pz = p.z;
    i = p.w;
  } else {
    i = order ? order[s] : s;
    px = a.xyz[3 * i];
    py = a.xyz[3 * i + 1];
    pz = a.xyz[3 * i + 2];
  }
  int4* r = srec + threadIdx.x * Q;
  const double* nsrc = pnorm ? pnorm + 3 * s : a.normals + 3 * i;
  const double nx = nsrc[0], ny = nsrc[1], nz = nsrc[2];
  r[0] = make_int4(px, py, pz, INF);
  r[1] = make_int4(__double2loint(nx), __double2hiint(nx), __double2loint(ny), __double2hiint(ny));
  r[2] = make_int4(__double2loint(nz), __double2hiint(nz), INF, 0);
  r[3] = make_int4(0, 0, 0, 0);
  // compact geometry (48 B per position) for the LDS tiles of static_mask_kernel; its spare word carries the
  // original index, so that the kernel can also tell which neighbours a point precedes
  sgeo[3 * threadIdx.x] = make_int4(px, py, pz, (int32_t)i);
  sgeo[3 * threadIdx.x + 1] = r[1];
  sgeo[3 * threadIdx.x + 2] = r[2];
  prio[s] = (int32_t)i;
  int row[KC];
#pragma unroll
  for (int j = 0; j < KC; j++) {
    int v = 0;
    if (j < a.K) {
      if (npos) {  // positions straight from the kNN kernels (rows in position order: coalesced)
        v = npos[s * a.K + j];
      } else {
        v = a.neigh[i * a.K + j];
        v = pos ? pos[v] : v;
      }
    }
    row[j] = v;
  }
  if (!npos) {
    // Caller-supplied rows may repeat an index in slots 1..K-1.  The reference labels a point on first sight and finds
    // it labelled on the second (my_function.cpp:226-233): the repeat is a no-op, and a seed with a repeat can never
    // collect K-1 accepted neighbours (:238).  The repeat becomes -1 = "no neighbour": the lane is idle in every Broad()
    // call, the static mask of the point is incomplete (never a plane seed), the first occurrence carries the rest.
    // (Our own k-lists hold distinct indices: the fused pipeline skips this.)
#pragma unroll
    for (int j = 2; j < KC; j++) {
      bool rep = false;
#pragma unroll
      for (int t = 1; t < j; t++)
        rep = rep || (j < a.K && row[t] == row[j]);
      row[j] = rep ? -1 : row[j];
    }
  }
#pragma unroll
  for (int j = 0; j < KC; j += 4)
    r[4 + j / 4] = make_int4(row[j], row[j + 1], row[j + 2], row[j + 3]);
  }  // s < n
  __syncthreads();
  const int64_t nv = a.n - b0 < 256 ? a.n - b0 : 256;
  int4* rout = rec + b0 * Q;
  for (int t = threadIdx.x; t < nv * Q; t += 256)
    rout[t] = srec[t];
  int4* gout = geo + 3 * b0;
  for (int t = threadIdx.x; t < nv * 3; t += 256)
    gout[t] = sgeo[t];
}

// per round: owner snapshot in, claims cleared
__global__ void refresh_records_kernel(const int32_t* __restrict__ omega, int4* __restrict__ rec, int quads, int64_t n)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  reinterpret_cast<int32_t*>(rec + i * quads)[3] = omega[i];
  *rec_tag(rec, quads, i) = INF;
}

// Plane state from the running sums (:249-250): normal = S / |S|, centre = (int32)((uint64)(int64)(int32)C / n).
// The three components sit in lanes 0..2, so ONE f64 division and ONE size_t division serve all of them (the same
// IEEE / integer operations on the same operands as three separate ones: same bits).  A free function on values, not
// part of the step's closure: written inside the lambda the closure fell out of registers (520 B of scratch per lane,
// every phase of the step 3x slower).
__device__ __forceinline__ void plane_state_lanes(int lane, double Sx, double Sy, double Sz, uint32_t Cx, uint32_t Cy, uint32_t Cz,
                                                  uint32_t n, double& cnx, double& cny, double& cnz, int& ccx, int& ccy, int& ccz)
{
  const double nrm = __builtin_sqrt((Sx * Sx) + (Sy * Sy) + (Sz * Sz));
  const double Sv = lane == 0 ? Sx : (lane == 1 ? Sy : Sz);
  const double qn = Sv / nrm;
  const CenterDiv cd = center_div_prepare(n);
  const int32_t Cv = lane == 0 ? (int32_t)Cx : (lane == 1 ? (int32_t)Cy : (int32_t)Cz);
  const int32_t qc = center_div(Cv, cd);
  cnx = readlane_f64(qn, 0);
  cny = readlane_f64(qn, 1);
  cnz = readlane_f64(qn, 2);
  ccx = readlane_i32(qc, 0);
  ccy = readlane_i32(qc, 1);
  ccz = readlane_i32(qc, 2);
}

constexpr int LDS_STACK = 256;  // LIFO entries (with rows) kept in LDS (128 entries = twice the workgroups per CU: 188.8 vs 182.6 ms at 50 M)
constexpr int LDS_REFILL = 128;  // entries brought back from HBM when the pops reach below the window
constexpr int MAX_RETRY_LONG = 3; // ... of which at most this many after a long list was thrown away
constexpr int MAX_RETRY = 12;    // in-launch re-growths of a plane that lost a point ...
constexpr int RETRY_MAX_LIST = 16384;  // ... as long as little work is thrown away (long planes wait for the next
                                      // round: their logged assumptions rarely survive their neighbours' insertion)

// Step engine.
//  * Every lane that gathers a neighbour gets that neighbour's own row with it
//    (same cache line); accepted children go on the LIFO with their rows (id in
//    slot 0).  The top LDS_STACK entries of the LIFO live in LDS (write-through to
//    the HBM slab, except the top entry, which the next call always consumes): a
//    pop is ONE LDS read per lane, and when the pops reach below the window it is
//    refilled from HBM (LDS_REFILL entries).  The kernel is bound by the
//    instruction issue of single waves (in-kernel cycle probes: the gather's
//    latency is fully hidden behind the plane-state arithmetic), so the hot loop
//    avoids per-lane select chains and everything else that costs scalar registers.
//  * Multi-pop: 64 % of all Broad() calls accept nothing, in long runs (DFS
//    backtracking).  A call that accepts nothing changes neither the plane
//    state nor any label, so the NEXT pending call sees exactly the same world:
//    the wave evaluates NG = 64 / KC pending calls at once (lane group g = the
//    g-th next call), consumes the leading run of empty ones, expands the first
//    non-empty one and leaves the rest on the LIFO.  Claims (atomics) are only
//    issued for the call that is really expanded.
template <int KC>
__global__ __launch_bounds__(64) void grow_spec_kernel(SpecArgs a, const unsigned long long* __restrict__ cand, int ncand,
                                                       int4* rec, int32_t* dead, Pool pool,
                                                       PlaneOut* __restrict__ out, int64_t step_cap, int retry_max_list,
                                                       const uint32_t* __restrict__ order)
{
  __shared__ __attribute__((aligned(16))) int lds_stack[LDS_STACK * KC];
  constexpr int Q = RecLayout<KC>::QUADS;
  constexpr int NG = 64 / KC;
  if ((int)blockIdx.x >= ncand)
    return;
  // attempts are numbered by seed (w), dispatched in the order the host chose (see dispatch_key_kernel)
  const int w = order ? (int)(order[blockIdx.x] & (MAX_WAVES - 1)) : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int g = lane / KC, j = lane % KC;
  const unsigned long long gmask0 = (KC == 32) ? 0xffffffffull : 0xffffull;
  const int K = a.K, nc = K - 1;
  const bool act = j < nc;
  const int64_t t_start = (int64_t)wall_clock64();
  const int32_t seed = (int32_t)(cand[w] >> 32);         // original index: what claims and owners are compared by
  const int32_t seed_s = (int32_t)(uint32_t)cand[w];     // position: where the seed's record lives
  Slab list = {0, 0}, stack = {0, 0}, log = {0, 0};
  int long_tries = 0;
  // 32-bit bookkeeping (n < 2^31): 64-bit scalar arithmetic doubles the SALU work of every call
  int ln = 1, sp = 0, lds_lo = 0, logn = 0;
  uint32_t iters = 0;
  const uint32_t iter_cap = step_cap > 0xFFFFFFF0ll ? 0xFFFFFFF0u : (uint32_t)step_cap;
  int status = ST_DONE;
  // Optimistic claims.  A claim is a NO-RETURN atomicMin on the record's tag (a returning atomic
  // keeps the wave waiting 600-950 cycles for a value that is rarely interesting).  What the
  // returned value used to tell is recovered without a wait:
  //  * "I took the point from a LATER plane": the tag read by this call's gather (tg > seed)
  //    names the victim, which is marked at claim time;
  //  * "an EARLIER plane got there first": the tag is read back by the NEXT call's gather
  //    (same wave, same address: the load observes the atomic) -- anything but my seed means lost.
  // A later plane that claims between my gather and my atomic is not marked by me, but reads the
  // tag back itself one call later and sees my seed.  Whatever slips through these advisory
  // checks is caught after the round: validate1 re-reads every list entry's tag and rejects
  // duplicated entries.
  bool pendv = false;
  const int32_t* vptr = dead;  // tag word claimed by the previous call (any valid address while !pendv)
  bool need_state = false;
  const int4* srec = rec + (int64_t)seed_s * Q;
  const int4 s0 = srec[0], s1 = srec[1], s2 = srec[2];
  double cnx, cny, cnz, Sx, Sy, Sz;
  int ccx, ccy, ccz;
  uint32_t Cx, Cy, Cz;
  // Deferred part of an expansion: the accepted points' normals and coordinates
  // (lanes d_am of the d_* registers) enter the running sums in list order, then
  // the plane's normal and centre follow.  It runs between the issue of the next
  // gather and the first use of its data, i.e. under the memory latency.
  unsigned long long d_am = 0;
  double d_mx = 0, d_my = 0, d_mz = 0;
  int d_px = 0, d_py = 0, d_pz = 0;
  auto update_state = [&]() {
    unsigned long long mm = d_am;
    while (mm) {
      const int l = __ffsll(mm) - 1;
      mm &= mm - 1;
      Sx += readlane_f64(d_mx, l);
      Sy += readlane_f64(d_my, l);
      Sz += readlane_f64(d_mz, l);
      Cx += (uint32_t)readlane_i32(d_px, l);
      Cy += (uint32_t)readlane_i32(d_py, l);
      Cz += (uint32_t)readlane_i32(d_pz, l);
    }
    plane_state_lanes(lane, Sx, Sy, Sz, Cx, Cy, Cz, (uint32_t)ln, cnx, cny, cnz, ccx, ccy, ccz);  // int /= size_t: quirk Q3, bs_centerdiv.h
    need_state = false;
  };
  // first slabs are small: nine attempts in ten fail at depth 0 and never need more
  const bool have_mem = slab_ensure(pool, list, 0, 256, lane) && slab_ensure(pool, stack, 0, 32 * (int64_t)KC, lane) &&
                        slab_ensure(pool, log, 0, 256, lane);
  // A plane that loses a point to a sequentially earlier plane is invalid, but the
  // earlier plane's claims are visible: instead of waiting for the next round the
  // wave releases its claims and grows the plane again at once, now treating the
  // earlier plane's points as taken (logged assumptions, validated after the round).
  // Chains of planes that depend on each other thus resolve inside ONE launch.
  for (int attempt = 0;; attempt++) {
  status = ST_DONE;
  pendv = false;
  ln = 1;
  sp = 0;
  lds_lo = 0;
  logn = 0;
  cnx = __hiloint2double(s1.y, s1.x);
  cny = __hiloint2double(s1.w, s1.z);
  cnz = __hiloint2double(s2.y, s2.x);
  ccx = s0.x;
  ccy = s0.y;
  ccz = s0.z;
  Sx = 0.0 + cnx;
  Sy = 0.0 + cny;
  Sz = 0.0 + cnz;
  Cx = (uint32_t)ccx;
  Cy = (uint32_t)ccy;
  Cz = (uint32_t)ccz;
  if (!have_mem) {
    status = ST_NOMEM;
  } else {
    if (lane == 0)
      pool.base[list.off] = seed_s;  // lists hold positions until they are committed
    // every pending call is a LIFO entry (id + row) in the LDS window; at the start
    // the only entry is Broad(seed, 0)
    if (lane < KC)
      lds_stack[lane] = reinterpret_cast<const int32_t*>(srec + 4)[lane];
    sp = 1;
    bool depth0 = true;
    int stack_entries = stack.cap / KC;  // LIFO entries the stack slab holds
    need_state = false;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // One Broad() step.  mode: 0 = no plane state pending, 1 = the previous step expanded (state to
    // be evaluated), 2 = decide at run time.  Returns 0 = consumed empty calls only, 1 = expanded a
    // call, 2 = leave (status set or LIFO empty).
    auto step = [&](auto mode) -> int {
      if (__builtin_expect(sp == 0, 0))
        return 2;
      if (__builtin_expect(++iters > iter_cap, 0)) {
        status = ST_WATCHDOG;
        return 2;
      }
      // ---- which pending call does my lane group evaluate? ----
      const int e = sp - 1 - g;                       // LIFO entry of my group
      const bool valid = e >= 0;
      const int ngv = (sp < NG) ? sp : NG;            // valid groups
      const int need_lo = (sp - NG > 0) ? sp - NG : 0;
      if (__builtin_expect(need_lo < lds_lo, 0)) {  // wave-uniform, rare
        const int new_lo = lds_lo > LDS_REFILL ? lds_lo - LDS_REFILL : 0;
        const int nw = (lds_lo - new_lo) * KC;
        for (int t = lane; t < nw; t += 64) {
          const int ee = new_lo + t / KC;
          lds_stack[(ee & (LDS_STACK - 1)) * KC + (t % KC)] = ld_i32(pool.base + stack.off + (int64_t)new_lo * KC + t);
        }
        lds_lo = new_lo;
      }
      int cand_id = 0;
      if (valid && act)
        cand_id = lds_stack[(e & (LDS_STACK - 1)) * KC + j + 1];
      const bool live = cand_id >= 0;  // (-1: repeated index of a caller-supplied row, no neighbour)
      cand_id = live ? cand_id : 0;
      const int killed = ld_i32(dead + seed);  // an earlier plane took one of my points: I am invalid
      const int vt = ld_i32(pendv ? vptr : dead + seed);  // read-back of the previous call's claim
      int own = 0, tg = INF, px = 0, py = 0, pz = 0;
      double mx = 0, my = 0, mz = 0;
      int row[KC];  // only read from lanes that loaded it (the accepting lanes)
      {
        const int4* r = rec + (int64_t)cand_id * Q;
        const int4 q0 = r[0], q1 = r[1], q2 = r[2];
        tg = ld_i32(reinterpret_cast<const int32_t*>(r + 2) + 2);
        px = q0.x;
        py = q0.y;
        pz = q0.z;
        own = q0.w;
        mx = __hiloint2double(q1.y, q1.x);
        my = __hiloint2double(q1.w, q1.z);
        mz = __hiloint2double(q2.y, q2.x);
#pragma unroll
        for (int t = 0; t < KC; t += 4) {
          const int4 v = r[4 + t / 4];
          row[t] = v.x;
          row[t + 1] = v.y;
          row[t + 2] = v.z;
          row[t + 3] = v.w;
        }
      }
      // The plane state of the previous expansion (square root, three f64 divisions,
      // three 64-bit integer divisions) is evaluated HERE, between the issue of the
      // gather loads above and their first use below: ~500 cycles of arithmetic
      // that hide behind the memory latency instead of preceding it.
      if constexpr (decltype(mode)::value == 1)
        update_state();
      else if constexpr (decltype(mode)::value == 2) {
        if (__builtin_expect(need_state, 1))
          update_state();
      }
      bool geo = false;
      if (valid && act && live && tg != seed) {  // tg == seed: already labelled by this plane
        const int dx = (int)((uint32_t)px - (uint32_t)ccx);
        const int dy = (int)((uint32_t)py - (uint32_t)ccy);
        const int dz = (int)((uint32_t)pz - (uint32_t)ccz);
        const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
        const double dt = cnx * mx + cny * my + cnz * mz;
        geo = dist <= a.th && dt >= a.cos_th;
      }
      // settle last call's optimistic claims: the tag must carry my seed
      const bool lost = pendv && vt != seed;  // an earlier plane got there first
      pendv = false;
      if (__builtin_expect(killed || ballot64(lost), 0)) {
        status = ST_STOLEN;
        return 2;
      }
      // side-effect free classification
      bool assume = geo && own < seed && !(own < a.F);  // kept by an earlier, not yet final attempt
      const bool contender = geo && !(own < seed);
      const unsigned long long cm = ballot64(contender);
      // ---- walk the pending calls in order: consume empty ones, stop at the first that accepts ----
      unsigned long long am = 0;
      int gstar = -1;
      bool ok = false;
      if (__builtin_expect(cm != 0, 1)) {  // cm == 0: every pending call is empty
        // first call with a contender; its contenders are normally all free or held by a later
        // plane (tag > seed): they simply claim, and the walk is over without a loop
        const int g1 = (__ffsll(cm) - 1) / KC;
        const unsigned long long gm1 = gmask0 << (g1 * KC);
        const unsigned long long earlier = ballot64(contender && tg < seed);  // held by an earlier in-flight plane
        if (__builtin_expect((earlier & gm1) == 0, 1)) {
          ok = contender && g == g1;
          if (ok) {
            int32_t* tp = rec_tag(rec, Q, cand_id);
            __hip_atomic_fetch_min(tp, seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // result unused: no-return form
            if (tg != INF)  // held by a later plane (tg > seed here): it is invalid now (value: thief + 1)
              __hip_atomic_store(dead + tg, seed + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vptr = tp;
            pendv = true;
          }
          am = cm & gm1;
          gstar = g1;
        } else {
        for (int gg = g1; gg < ngv; gg++) {
          const unsigned long long gm = gmask0 << (gg * KC);
          if (cm & gm) {
            // Claim protocol for call gg.  tag[p] = seed of the in-flight plane
            // holding p.  A sequentially earlier plane always wins (atomicMin); a
            // plane that loses a point it had accepted is marked dead, and points
            // held by dead planes can be reclaimed.  Every "taken" decision that
            // rests on a non-final owner is logged and re-checked after the round.
            if (g == gg && contender) {
              int32_t* tp = rec_tag(rec, Q, cand_id);
              int cur_tag = tg;
              bool mine = false;
              for (int tries = 0; tries < 8 && !ok && !assume && !mine; tries++) {
                if (cur_tag < seed) {            // an earlier in-flight plane holds it ...
                  if (ld_i32(dead + cur_tag) < 0) {  // ... which has given up for this launch: reclaim
                    const int old = atomicCAS(tp, cur_tag, seed);
                    if (old == cur_tag)
                      ok = true;
                    else
                      cur_tag = old;
                  } else {
                    assume = true;
                  }
                } else if (cur_tag == seed) {
                  mine = true;
                } else {  // free, or held by a later plane: claim (rare path: returning atomic, settled at once)
                  const int old = atomicMin(tp, seed);
                  if (old > seed) {
                    if (old != INF)
                      __hip_atomic_store(dead + old, seed + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = true;
                  } else if (old == seed) {
                    mine = true;
                  } else {
                    cur_tag = old;  // an earlier plane got there between the gather and the claim: look again
                  }
                }
              }
              if (!ok && !mine)
                assume = true;
            }
            am = ballot64(ok);
            if (am) {
              gstar = gg;
              break;
            }
          }
        }
        }
      }
      const int last = gstar >= 0 ? gstar : ngv - 1;  // calls 0..last are consumed
      // assumptions made by consumed calls
      const unsigned long long lm = ballot64(assume && g <= last);
      const int lcnt = __popcll(lm);
      if (__builtin_expect(lcnt != 0, 0)) {
        if (!slab_ensure(pool, log, logn, logn + lcnt, lane)) {
          status = ST_NOMEM;
          return 2;
        }
        if (assume && g <= last)
          pool.base[log.off + logn + __popcll(lm & ((1ull << lane) - 1ull))] = cand_id;
        logn += lcnt;
      }
      const int cnt = __popcll(am);
      if (__builtin_expect(depth0 && cnt < nc, 0)) {
        status = ST_FAILED0;  // under speculation this seed is (currently) an orphan maker
        return 2;
      }
      depth0 = false;
      // pops: every consumed stack-fed call (the expanded one included)
      sp -= last + 1;
      if (sp < lds_lo)
        lds_lo = sp;  // entries below are only in HBM; new pushes land in LDS again
      if (__builtin_expect(gstar < 0, 0))
        return 0;
      // ---- expand call gstar: :231-255 ----
      if (__builtin_expect(ln + cnt > list.cap || sp + cnt > stack_entries, 0)) {  // one test for both slabs; growing them is rare
        if (!slab_ensure(pool, list, ln, ln + cnt, lane) ||
            !slab_ensure(pool, stack, sp > 0x7ffffff0 / KC ? 0x7ffffff0 : sp * KC, (int64_t)(sp + cnt) * KC, lane)) {
          status = ST_NOMEM;
          return 2;
        }
        stack_entries = stack.cap / KC;
      }
      const int rank = __popcll(am & ((1ull << lane) - 1ull));
      if (ok)
        pool.base[list.off + ln + rank] = cand_id;
      // the running sums (:231-248) and the plane state (:249-250) are evaluated
      // after the next gather has been issued: keep what they need
      d_am = am;
      d_mx = mx;
      d_my = my;
      d_mz = mz;
      d_px = px;
      d_py = py;
      d_pz = pz;
      ln += cnt;
      need_state = true;
      // the children go on the LIFO (reversed) with their rows; id in slot 0.  The
      // first child (top entry) is consumed by the very next call and therefore
      // never needs its write-through copy in HBM.
      if (ok) {
        const int pe = sp + (cnt - 1 - rank);
        int4* lslot = reinterpret_cast<int4*>(lds_stack + (pe & (LDS_STACK - 1)) * KC);
        row[0] = cand_id;
#pragma unroll
        for (int t = 0; t < KC; t += 4)
          lslot[t / 4] = make_int4(row[t], row[t + 1], row[t + 2], row[t + 3]);
        if (rank > 0) {
          int4* slot = reinterpret_cast<int4*>(pool.base + stack.off + (int64_t)pe * KC);
#pragma unroll
          for (int t = 0; t < KC; t += 4)
            slot[t / 4] = make_int4(row[t], row[t + 1], row[t + 2], row[t + 3]);
        }
      }
      sp += cnt;
      if (__builtin_expect(sp - lds_lo > LDS_STACK, 0))
        lds_lo = sp - LDS_STACK;  // older entries were overwritten in LDS (still in HBM)
      return 1;
    };
    if constexpr (KC == 32) {
      // Nested loops over per-case instances of the step: the plane state (sums, normal, centre)
      // is redefined at ONE place per loop and is invariant in the inner loop of empty steps, which
      // spares the merge copies at the back edge (10 M, k=32: -6 % kernel time; the k<=16 build is
      // 7 % FASTER with the single loop below).
      int r = step(std::integral_constant<int, 0>{});
      while (r == 0)
        r = step(std::integral_constant<int, 0>{});
      while (r == 1) {
        r = step(std::integral_constant<int, 1>{});
        while (r == 0)
          r = step(std::integral_constant<int, 0>{});
      }
    } else {
      while (step(std::integral_constant<int, 2>{}) != 2) {
      }
    }
  }
  if (need_state)  // state of the very last expansion (the plane's reported normal / centre)
    update_state();
  // Read back the claims of the very last call (victims were marked at claim time on every path).
  if (status == ST_DONE) {
    const bool lost = pendv && ld_i32(vptr) != seed;
    if (ballot64(lost))
      status = ST_STOLEN;
  }
  // (a LONG plane -- more than RETRY_MAX_LIST entries thrown away -- gets at most MAX_RETRY_LONG further tries)
  bool seed_taken = false;
  if (status == ST_STOLEN) {  // my own seed point claimed by an earlier live plane: in the reference this seed never attempts
    const int t0 = ld_i32(rec_tag(rec, Q, seed_s));
    seed_taken = t0 < seed && ld_i32(dead + t0) >= 0;
  }
  if (status != ST_STOLEN || seed_taken || attempt >= MAX_RETRY || ln > retry_max_list ||
      (ln > RETRY_MAX_LIST && ++long_tries > MAX_RETRY_LONG))
    break;
  // release this incarnation's claims (points taken over by others keep their new tag) ...
  for (int t = 1 + lane; t < ln; t += 64)
    atomicCAS(rec_tag(rec, Q, ld_i32(pool.base + list.off + t)), seed, INF);
  // ... and come back to life: marks written from here on concern the new incarnation
  __hip_atomic_store(dead + seed, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }  // attempt loop
  // dead[] protocol.  > 0: "a thief (value - 1) took one of my points" -- written by thieves,
  // read by the victim, which may come back to life in this launch (re-growth above) and claim
  // the very same points again under the very same tag value.  Reclaiming from such a plane
  // is an ABA race (read the flag, the victim resurrects and re-claims, then the CAS on the
  // tag succeeds against the LIVE incarnation, which later takes the point back and holds
  // it twice -- found by fuzzing as a validated plane with duplicated list entries).  Only a
  // plane that has given up for this launch is reclaimable: it says so itself, here, with
  // a negative value, and never claims again.
  if (lane == 0 && status == ST_STOLEN) {
    const int t = ld_i32(dead + seed);
    __hip_atomic_store(dead + seed, t > 0 ? -t : (t == 0 ? -0x40000000 : t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    o.steps = iters;  // wave steps (a multi-pop counts once)
    o.seed = seed;
    o.status = status;
    o.keep = (status == ST_DONE && ln > a.th_count) ? 1 : 0;
    o.consistent = 0;
    o.pad = 0;
    const int dflag = ld_i32(dead + seed);
    o.thief = (dflag < 0 ? -dflag : dflag) - 1;
    o.seed_pos = seed_s;
    o.v3ok = 1;
    o.pad4 = 0;
    o.t_start = t_start;
    o.t_end = (int64_t)wall_clock64();
    o.w = w;
    // where the wave ran (debug print only): HW_ID bits [15:0] (wave, SIMD, CU, SH, SE) | XCC_ID << 16
    o.pad5 = (int32_t)((__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xffff) |
                       ((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf) << 16));
    out[w] = o;
  }
}

// ---- step engine, second generation (BS_GROW_V2=0 selects the kernel above) -------------------------------------
// Same protocol, same arithmetic, same results as grow_spec_kernel -- the step is re-cut around what the stamps of
// the first engine showed (~3 000 cycles per step of ONE wave; identical at 19.6 k-, 90 k- and 250 k-point planes,
// i.e. whether the records fit an XCD's L2 or not: the step is bound by the issue of the wave's own instruction
// stream, not by where the gather is served from).  What changed:
//  * HOT LOOP + COMPLETE STEP.  The loop that runs per step contains the common step only: pending calls in the LDS
//    window, the first call with contenders claims without meeting an earlier plane, nothing to log, room in the LDS
//    rings.  Everything else (refill, slow claim walk, assumption log, ring flushes, the depth-0 rule) is noticed
//    BEFORE the step has a side effect and handled by ONE run of the complete step outside the loop.  With the rare
//    paths inlined in the loop body the compiler kept its loop-carried scalars per lane in vector registers under
//    exec-mask control flow (190-208 VGPRs, 57 spilled SGPRs); uniformity is also said explicitly (readfirstlane)
//    where the divergence analysis gives up (merged exits behind per-lane branches, per-lane trip counts);
//  * every vector-memory instruction of the hot loop is inline assembly behind ONE explicit wait: the claim tag rides
//    in the same 16-byte sc1 load as the normal's third component, and no compiler-placed s_waitcnt vmcnt(0) sits
//    right behind the previous step's claim atomics (600-900 cycles);
//  * the LIFO lives in LDS only: its oldest 128 entries are flushed to the HBM slab in one coalesced burst when the
//    256-entry window is full and come back the same way -- no write-through per push;
//  * pointIdx entries are collected in an LDS ring and leave in coalesced bursts: no store, no address arithmetic
//    and no slab-capacity check in the step;
//  * distance and normal test are evaluated side by side (no short circuit);
//  * the gather is waited for in two parts: the five loads the test needs (flags, position, normal, tag) first, the
//    candidates' own rows -- issued last, needed by the push only -- after the classification.  Every scattered
//    64-lane load costs ~70 cycles of the CU's address unit whatever it hits (tools/probe/gather_probe.hip: 369 cycles
//    for one, 805 for seven), so the four row loads used to hold the test back by ~280 cycles (facade 131 -> 121 ms).
//    The rows' registers are written by the hardware after the asm statement that names them: there is ONE wait for
//    them on every path, and tests/test_isa_rows_wait.py checks in the generated ISA that nothing touches them before.
// Tried and measured slower (facade 1 M, growth kernels; at the time this engine took 136.6 ms, later 131.2 ms, the
// first engine 147.5 ms):
//  * the candidates' neighbour rows fetched cooperatively (four lanes per row, 16 cache lines per instruction
//    instead of 60, rows kept in that layout until the LDS push): 150-159 ms -- the four extra LIFO reads for the
//    row duty and the issue of the re-addressed loads cost more than the saved line look-ups;
//  * a single-exit loop (decide `commit` first, apply under one branch): 146.9 ms;
//  * an LDS bitmap of the points this plane already holds (no record fetched for them: their lanes share one cache
//    line; a step that only sees own points needs no memory access): 145.6 ms -- the time from the issue of the gather
//    to its data (~1 150 cycles together with the plane-state arithmetic under it) does not depend on how many lanes
//    fetch distinct lines;
//  * the claim tags in an array of their own instead of inside the 128-byte records (so that the claims' memory-side
//    atomics would not throw the records' lines out of L2): first engine 149.3 vs 147.8 ms -- neutral.
//  * the flag / read-back / record loads under exec masks (lane 0 alone reads the killed flag, only claimers read
//    back, only testing lanes fetch records: ~110 instead of ~550 lane requests per step): the gather came back ~90
//    cycles earlier, but the three masked regions and the defined-before-asm registers cost 440 cycles at the issue:
//    146.9 ms;
//  * one combined test per kind of loop exit instead of separate breaks: 145.0 ms (the back edge grew to 570 cycles).
//  * the plane state on a second wave of the workgroup (128 threads: wave 1 keeps the running sums and publishes
//    normal and centre through LDS, wave 0 traverses).  With a release/acquire hand-off 143.4 vs 131.2 ms (the
//    release waits for the claim atomics' round trip); with a fence-free sequence number in LDS (the LDS operations
//    of a wave are executed in order) and the accepted points posted before the claims: facade 129.3 vs 120.6 ms,
//    urban 10 M 106.6 vs 111.5 ms.  The stamps say why: the state wave needs 1 270 cycles per expansion and is never
//    late, but wave 0 now waits 757 instead of 147 cycles for its gather -- on a 128 MB record array the gather
//    chain (pop 150 + issue 350 + Infinity Cache 850) is what the arithmetic used to hide under, so taking the
//    arithmetic away buys nothing there, and the posting costs 230 cycles;
//  * whole records fetched cooperatively into LDS (global_load_lds_dwordx4, eight lanes per 128-byte record, eight
//    instructions per 64 candidates): in the micro-benchmark 920-1 000 cycles against 890 for the nine per-lane loads
//    with one wave per CU -- it only pays at 8+ waves per CU (3 200 against 4 600), where the launches are not bound;
//  * s_setprio 3 for attempts past 512 steps (the launch ends with its longest plane): urban 10 M 113.0 vs 108.7 ms,
//    50 M 186.5 vs 181.4 ms;
//  * two steps per trip of the hot loop (the back edge is a cascade of exit-flag blocks, ~35 scalar instructions and
//    five taken branches): facade 122.6 vs 120.6-123 ms -- nothing.
// What bounds a step (stamps of this engine on facade 1 M, cycles: pop + issue 510, plane state 700 over it, wait 150,
// test + classification 265, claims 390, list ring 305, push 150, loop control and exits 360 = 2 900) is the wave's own
// instruction stream -- one wave issues one instruction every ~6 cycles, and every phase consumes what the one before
// produced -- around a gather chain of ~1 350 cycles that the state arithmetic covers.  Issuing the next gather
// before this step's claims would overlap the two halves, but the next step's candidates are the neighbours of the
// points claimed now: the tags they read must already carry the claims, or a point is accepted twice.
constexpr int LBUF = 256;  // LDS ring of pointIdx entries (flushed in bursts of 64)
#ifdef BS_PROBE
// cycle stamps between the phases of a step (developer build: tools/probe_grow2.sh); planes with > 20 000 entries
__device__ unsigned long long g_prof2[32];
#define PROBE(i)                    \
  do {                              \
    const long long _t = clock64(); \
    pacc[i] += _t - tlast;          \
    tlast = _t;                     \
  } while (0)
#else
#define PROBE(i) \
  do {           \
  } while (0)
#endif

typedef int v4i __attribute__((ext_vector_type(4)));  // a native 128-bit register tuple (inline-asm operand)

template <int KC>
__global__ __launch_bounds__(64) void grow_spec2_kernel(SpecArgs a, const unsigned long long* __restrict__ cand, int ncand,
                                                        int4* rec, int32_t* dead, Pool pool,
                                                        PlaneOut* __restrict__ out, int64_t step_cap, int retry_max_list,
                                                        const uint32_t* __restrict__ order)
{
  __shared__ __attribute__((aligned(16))) int lds_stack[LDS_STACK * KC];
  __shared__ int lbuf[LBUF];
  constexpr int Q = RecLayout<KC>::QUADS;
  constexpr int NG = 64 / KC;        // pending calls evaluated per step
  constexpr int CH = KC / 4;         // 16-byte chunks of a neighbour row
  if ((int)blockIdx.x >= ncand)
    return;
  const int w = order ? (int)(order[blockIdx.x] & (MAX_WAVES - 1)) : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int g = lane / KC, j = lane % KC;
  const unsigned long long gmask0 = (KC == 32) ? 0xffffffffull : 0xffffull;
  const int K = a.K, nc = K - 1;
  const bool act = j < nc;
  const int64_t t_start = (int64_t)wall_clock64();
  const int32_t seed = (int32_t)(cand[w] >> 32);
  const int32_t seed_s = (int32_t)(uint32_t)cand[w];
  Slab list = {0, 0}, stack = {0, 0}, log = {0, 0};
  int long_tries = 0;
  int ln = 1, lflushed = 0, sp = 0, lds_lo = 0, logn = 0;
  uint32_t iters = 0;
  const uint32_t iter_cap = step_cap > 0xFFFFFFF0ll ? 0xFFFFFFF0u : (uint32_t)step_cap;
  int status = ST_DONE;
  bool pendv = false;
  const int32_t* vptr = dead;
  bool need_state = false;
  const int4* srec = rec + (int64_t)seed_s * Q;
  const int4 s0 = srec[0], s1 = srec[1], s2 = srec[2];
  double cnx, cny, cnz, Sx, Sy, Sz;
  int ccx, ccy, ccz;
  uint32_t Cx, Cy, Cz;
  unsigned long long d_am = 0;
  double d_mx = 0, d_my = 0, d_mz = 0;
  int d_px = 0, d_py = 0, d_pz = 0;
  const bool have_mem = slab_ensure(pool, list, 0, 256, lane) && slab_ensure(pool, stack, 0, 32 * (int64_t)KC, lane) &&
                        slab_ensure(pool, log, 0, 256, lane);
  // pointIdx entries [lflushed, ln) are in the LDS ring; everything below is in the HBM slab
  auto flush_list = [&](int upto) -> bool {  // make entries [lflushed, upto) durable (upto <= ln)
    if (upto <= lflushed)
      return true;
    if (!slab_ensure(pool, list, lflushed, upto, lane))
      return false;
    for (int t0 = lflushed; t0 < upto; t0 += 64) {  // (uniform trip counts throughout: see slab_ensure)
      const int t = t0 + lane;
      if (t < upto)
        pool.base[list.off + t] = lbuf[t & (LBUF - 1)];
    }
    lflushed = upto;
    return true;
  };
  for (int attempt = 0;; attempt++) {
    status = ST_DONE;
    pendv = false;
    ln = 1;
    lflushed = 0;
    sp = 0;
    lds_lo = 0;
    logn = 0;
    cnx = __hiloint2double(s1.y, s1.x);
    cny = __hiloint2double(s1.w, s1.z);
    cnz = __hiloint2double(s2.y, s2.x);
    ccx = s0.x;
    ccy = s0.y;
    ccz = s0.z;
    Sx = 0.0 + cnx;
    Sy = 0.0 + cny;
    Sz = 0.0 + cnz;
    Cx = (uint32_t)ccx;
    Cy = (uint32_t)ccy;
    Cz = (uint32_t)ccz;
    need_state = false;
    if (!have_mem) {
      status = ST_NOMEM;
    } else {
      if (lane == 0)
        lbuf[0] = seed_s;  // lists hold positions until they are committed
      if (lane < KC)
        lds_stack[lane] = reinterpret_cast<const int32_t*>(srec + 4)[lane];
      sp = 1;
      bool depth0 = true;
      __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef BS_PROBE
      long long pacc[12] = {0};
      long long ncalls = 0, nexp = 0;
      long long tlast = clock64();
#endif
      // ---- what every step starts with: pop, gather, plane state of the previous expansion, test, settle ----
      // (straight-line code without exits: it is shared by the hot loop and by the complete step)
      struct Ev {
        int cand_id, px, py, pz, own, tg;
        double mx, my, mz;
        bool geo, valid;
        int ngv;
        int killed_v, vt;  // (in flight until rows_wait)
        bool lost_any;  // killed, or a claim of the previous call was lost
        v4i rows[CH];  // the candidate's own neighbour row (it becomes its LIFO entry if it is accepted)
      };
      auto eval = [&](Ev& E) {
        E.ngv = (sp < NG) ? sp : NG;
        const int e = sp - 1 - g;
        E.valid = e >= 0;
        // unconditional LDS reads (the index is masked into the ring, the value is discarded where it means nothing):
        // a read under `if` costs an exec-mask branch each, five of them in a row at the head of every step
        {
          const int v = lds_stack[(e & (LDS_STACK - 1)) * KC + ((j + 1) & (KC - 1))];
          E.valid = E.valid && v >= 0;  // (-1: repeated index of a caller-supplied row, no neighbour)
          E.cand_id = (E.valid && act) ? v : 0;
        }
        // EVERY vector-memory instruction of the hot loop is issued from inline assembly and waited for by the ONE
        // explicit s_waitcnt below.  Left to the compiler, the loop's first re-use of a load's destination register
        // gets an s_waitcnt vmcnt(0) (the wait-count pass merges the back edge's pending events conservatively), and
        // that wait sits right behind the previous step's claim atomics: 600-900 cycles per step.
        // (the killed flag is read by every lane from ONE address, but the compiler cannot know that the lanes agree)
        v4i q0, q1, q2;
        {
          const int32_t* kp = dead + seed;
          const int32_t* vp = pendv ? vptr : dead + seed;
          const int4* r = rec + (int64_t)E.cand_id * Q;
          // what the test needs first, then what only the commit needs: the killed flag, the read-back of the previous
          // claims and (below) the candidates' rows
          asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                       "global_load_dwordx4 %1, %5, off offset:16\n\t"
                       "global_load_dwordx4 %2, %5, off offset:32 sc1\n\t"
                       "global_load_dword %3, %6, off sc1\n\t"
                       "global_load_dword %4, %7, off sc1"
                       : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(E.killed_v), "=&v"(E.vt)
                       : "v"(r), "v"(kp), "v"(vp)
                       : "memory");
#pragma unroll
          for (int t = 0; t < CH; t++) {
            const int4* rp = r + 4 + t;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(E.rows[t]) : "v"(rp) : "memory");
          }
        }
        PROBE(0);
        // The plane state of the previous expansion is evaluated HERE, under the gather's latency.
        if (__builtin_expect(need_state, 1)) {
          unsigned long long mm = d_am;
          while (mm) {
            const int l = __ffsll(mm) - 1;
            mm &= mm - 1;
            Sx += readlane_f64(d_mx, l);
            Sy += readlane_f64(d_my, l);
            Sz += readlane_f64(d_mz, l);
            Cx += (uint32_t)readlane_i32(d_px, l);
            Cy += (uint32_t)readlane_i32(d_py, l);
            Cz += (uint32_t)readlane_i32(d_pz, l);
          }
          plane_state_lanes(lane, Sx, Sy, Sz, Cx, Cy, Cz, (uint32_t)ln, cnx, cny, cnz, ccx, ccy, ccz);
          need_state = false;
        }
        PROBE(1);
        // ---- the ONE wait of the gather (the operands tie every later use of the data behind it) ----
        // (loads return in issue order: with CH + 2 still outstanding the three record loads have landed; the flags and
        // the candidates' own rows are only needed by the commit and keep streaming in under the test: rows_wait() below)
        if (CH == 4)
          asm volatile("s_waitcnt vmcnt(6)" : "+v"(q0), "+v"(q1), "+v"(q2)::"memory");
        else
          asm volatile("s_waitcnt vmcnt(10)" : "+v"(q0), "+v"(q1), "+v"(q2)::"memory");
        PROBE(2);
        E.px = q0.x;
        E.py = q0.y;
        E.pz = q0.z;
        E.own = q0.w;
        E.tg = q2.z;
        E.mx = __hiloint2double(q1.y, q1.x);
        E.my = __hiloint2double(q1.w, q1.z);
        E.mz = __hiloint2double(q2.y, q2.x);
        {
          const int dx = (int)((uint32_t)E.px - (uint32_t)ccx);
          const int dy = (int)((uint32_t)E.py - (uint32_t)ccy);
          const int dz = (int)((uint32_t)E.pz - (uint32_t)ccz);
          const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
          const double dt = cnx * E.mx + cny * E.my + cnz * E.mz;
          const bool okd = dist <= a.th, okn = dt >= a.cos_th;      // both chains run side by side
          E.geo = E.valid & act & (E.tg != seed) & okd & okn;       // tg == seed: already labelled by this plane
        }
      };
      // The rows of the candidates are still in flight after eval(): EVERY path from eval() on passes through this wait
      // before it reads them, issues another vector-memory instruction, or leaves the step (their destination registers
      // must stay allocated until the loads have landed).
      auto rows_wait = [&](Ev& E) {
        static_assert(CH == 4 || CH == 8, "row chunks");
        if constexpr (CH == 4)
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(E.rows[0]), "+v"(E.rows[1]), "+v"(E.rows[2]), "+v"(E.rows[3]), "+v"(E.killed_v), "+v"(E.vt)::"memory");
        else
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(E.rows[0]), "+v"(E.rows[1]), "+v"(E.rows[2]), "+v"(E.rows[3]), "+v"(E.rows[4]), "+v"(E.rows[5]),
                         "+v"(E.rows[6]), "+v"(E.rows[7]), "+v"(E.killed_v), "+v"(E.vt)::"memory");
        const bool lost = pendv && E.vt != seed;  // an earlier plane got there first
        E.lost_any = __builtin_amdgcn_readfirstlane(E.killed_v) != 0 || ballot64(lost) != 0;
      };
      // ---- expansion of call gstar (:231-255): list ring, deferred state, children onto the LIFO ----
      auto expand = [&](const Ev& E, bool ok, unsigned long long am, int gstar, int cnt) {
        const int rank = __popcll(am & ((1ull << lane) - 1ull));
        if (ok)
          lbuf[(ln + rank) & (LBUF - 1)] = E.cand_id;
        d_am = am;
        d_mx = E.mx;
        d_my = E.my;
        d_mz = E.mz;
        d_px = E.px;
        d_py = E.py;
        d_pz = E.pz;
        ln += cnt;
        need_state = true;
        PROBE(7);
        // children onto the LIFO (reversed) with their rows; id in slot 0
        if (ok) {
          const int pe = sp + (cnt - 1 - rank);
          v4i* lslot = reinterpret_cast<v4i*>(lds_stack + (pe & (LDS_STACK - 1)) * KC);
#pragma unroll
          for (int t = 0; t < CH; t++) {
            v4i v = E.rows[t];
            if (t == 0)
              v.x = E.cand_id;
            lslot[t] = v;
          }
        }
        (void)gstar;
        sp += cnt;
      };
      // ---- the COMPLETE step: every rare path (refill, slow claim walk, assumption log, ring flushes, depth 0) ----
      // Returns 0 = step done, 2 = leave (status set or LIFO empty).  It is kept OUT of the hot loop's body: with the
      // rare paths inlined there the loop-carried scalars were spilled or kept per lane (57 spilled SGPRs, 190-208 VGPRs).
      auto step_complete = [&]() -> int {
        if (__builtin_expect(sp == 0, 0))
          return 2;
        if (__builtin_expect(++iters > iter_cap, 0)) {
          status = ST_WATCHDOG;
          return 2;
        }
        const int need_lo = (sp - NG > 0) ? sp - NG : 0;
        if (__builtin_expect(need_lo < lds_lo, 0)) {  // the pops reach below the LDS window: bring LDS_REFILL entries back
          const int new_lo = lds_lo > LDS_REFILL ? lds_lo - LDS_REFILL : 0;
          const int nw = (lds_lo - new_lo) * KC;
          for (int t0 = 0; t0 < nw; t0 += 64) {
            const int t = t0 + lane;
            const int ee = new_lo + t / KC;
            if (t < nw)
              lds_stack[(ee & (LDS_STACK - 1)) * KC + (t % KC)] = ld_i32(pool.base + stack.off + (int64_t)new_lo * KC + t);
          }
          lds_lo = new_lo;
        }
        Ev E;
        eval(E);
        rows_wait(E);
        pendv = false;
        if (__builtin_expect(E.lost_any, 0)) {
          status = ST_STOLEN;
          return 2;
        }
        bool assume = E.geo && E.own < seed && !(E.own < a.F);
        const bool contender = E.geo && !(E.own < seed);
        const unsigned long long cm = ballot64(contender);
        unsigned long long am = 0;
        int gstar = -1;
        bool ok = false;
        const int cand_id = E.cand_id, tg = E.tg, ngv = E.ngv;
        if (cm != 0) {
          const int g1 = (__ffsll(cm) - 1) / KC;
          const unsigned long long gm1 = gmask0 << (g1 * KC);
          const unsigned long long earlier = ballot64(contender && tg < seed);
          if ((earlier & gm1) == 0) {
            ok = contender && g == g1;
            if (ok) {
              int32_t* tp = rec_tag(rec, Q, cand_id);
              __hip_atomic_fetch_min(tp, seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // no-return form
              if (tg != INF)
                __hip_atomic_store(dead + tg, seed + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              vptr = tp;
              pendv = true;
            }
            am = cm & gm1;
            gstar = g1;
          } else {
            for (int gg = g1; gg < ngv; gg++) {
              const unsigned long long gm = gmask0 << (gg * KC);
              if (cm & gm) {
                if (g == gg && contender) {
                  int32_t* tp = rec_tag(rec, Q, cand_id);
                  int cur_tag = tg;
                  bool mine = false;
                  for (int tries = 0; tries < 8 && !ok && !assume && !mine; tries++) {
                    if (cur_tag < seed) {
                      if (ld_i32(dead + cur_tag) < 0) {
                        const int old = atomicCAS(tp, cur_tag, seed);
                        if (old == cur_tag)
                          ok = true;
                        else
                          cur_tag = old;
                      } else {
                        assume = true;
                      }
                    } else if (cur_tag == seed) {
                      mine = true;
                    } else {
                      const int old = atomicMin(tp, seed);
                      if (old > seed) {
                        if (old != INF)
                          __hip_atomic_store(dead + old, seed + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = true;
                      } else if (old == seed) {
                        mine = true;
                      } else {
                        cur_tag = old;
                      }
                    }
                  }
                  if (!ok && !mine)
                    assume = true;
                }
                am = ballot64(ok);
                if (am) {
                  gstar = gg;
                  break;
                }
              }
            }
          }
        }
        const int last = gstar >= 0 ? gstar : ngv - 1;
        const unsigned long long lm = ballot64(assume && g <= last);
        if (lm != 0) {
          const int lcnt = __popcll(lm);
          if (!slab_ensure(pool, log, logn, logn + lcnt, lane)) {
            status = ST_NOMEM;
            return 2;
          }
          if (assume && g <= last)
            pool.base[log.off + logn + __popcll(lm & ((1ull << lane) - 1ull))] = cand_id;
          logn += lcnt;
        }
        const int cnt = __popcll(am);
        if (depth0 && cnt < nc) {
          status = ST_FAILED0;
          return 2;
        }
        depth0 = false;
        sp -= last + 1;
        if (gstar < 0)
          return 0;
        if (ln + cnt - lflushed > LBUF) {  // the ring is full: a coalesced burst to the HBM slab
          if (!flush_list(ln)) {
            status = ST_NOMEM;
            return 2;
          }
        }
        if (sp + cnt - lds_lo > LDS_STACK) {  // LIFO window full: its oldest half goes to HBM
          const int upto = lds_lo + LDS_STACK / 2;
          if (!slab_ensure(pool, stack, lds_lo > 0x7ffffff0 / KC ? 0x7ffffff0 : lds_lo * KC, (int64_t)upto * KC, lane)) {
            status = ST_NOMEM;
            return 2;
          }
          for (int t0 = 0; t0 < (LDS_STACK / 2) * KC; t0 += 64) {
            const int t = t0 + lane;
            const int ee = lds_lo + t / KC;
            pool.base[stack.off + (int64_t)lds_lo * KC + t] = lds_stack[(ee & (LDS_STACK - 1)) * KC + (t % KC)];
          }
          lds_lo = upto;
        }
        expand(E, ok, am, gstar, cnt);
        return 0;
      };
      // (what a complete step leaves behind is wave-uniform; said explicitly, because its exits merge behind per-lane
      // branches and the divergence analysis gives up on them -- and then on every scalar of the hot loop)
      auto uniform_state = [&]() {
        sp = __builtin_amdgcn_readfirstlane(sp);
        ln = __builtin_amdgcn_readfirstlane(ln);
        lds_lo = __builtin_amdgcn_readfirstlane(lds_lo);
        lflushed = __builtin_amdgcn_readfirstlane(lflushed);
        logn = __builtin_amdgcn_readfirstlane(logn);
        iters = (uint32_t)__builtin_amdgcn_readfirstlane((int)iters);
        status = __builtin_amdgcn_readfirstlane(status);
        d_am = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(d_am >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)d_am);
      };
      // ---- driver: a complete step (the first one is the depth-0 step), then the hot loop until a step is not its kind ----
      for (;;) {
        const int r = __builtin_amdgcn_readfirstlane(step_complete());
        uniform_state();
        if (r == 2)
          break;
        bool leave = false;
        // (nothing may be in flight when the hot loop is entered: a load still pending at its preheader makes the
        // wait-count pass put an s_waitcnt vmcnt(0) at the loop's first register re-use -- executed in EVERY iteration,
        // where it waits for the previous step's claim atomics, 600-900 cycles)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // ---- the HOT LOOP: the common step only, every exit is a wave-uniform break BEFORE the step has a side effect ----
        for (;;) {
          if (__builtin_expect(sp == 0, 0)) {
            leave = true;
            break;
          }
          PROBE(9);
#ifdef BS_PROBE
          ncalls++;
#endif
          if (__builtin_expect(iters >= iter_cap, 0))
            break;  // (the complete step raises the watchdog)
          if (__builtin_expect(((sp - NG > 0) ? sp - NG : 0) < lds_lo, 0))
            break;  // refill
          Ev E;
          eval(E);
          // Classification first, ONE rows_wait() for every way on (a wait per exit made the compiler copy the row
          // registers -- still being written by the loads -- where the paths meet), then the exits.
          const bool assume = E.geo && E.own < seed && !(E.own < a.F);
          const bool contender = E.geo && !(E.own < seed);
          const unsigned long long cm = ballot64(contender);
          PROBE(3);
          int gstar = -1, cnt = 0;
          unsigned long long am = 0;
          bool slow = false;
          if (__builtin_expect(cm != 0, 1)) {
            gstar = (__ffsll(cm) - 1) / KC;
            const unsigned long long gm1 = gmask0 << (gstar * KC);
            am = cm & gm1;
            cnt = __popcll(am);
            // everything this loop cannot do is known here, before the first claim is issued
            slow = (ballot64(contender && E.tg < seed) & gm1) != 0 || ballot64(assume && g <= gstar) != 0 ||
                   ln + cnt - lflushed > LBUF || sp - (gstar + 1) + cnt - lds_lo > LDS_STACK;
          } else {
            slow = ballot64(assume) != 0;  // only empty calls, but one of them assumed something: the complete step logs it
          }
          rows_wait(E);  // (before the claims: a wait behind them would put the atomics' round trip on the chain)
          if (__builtin_expect(E.lost_any, 0)) {
            pendv = false;
            status = ST_STOLEN;
            leave = true;
            break;
          }
          if (__builtin_expect(slow, 0))
            break;
          // ---- committed: from here on the step has side effects ----
          iters++;
          pendv = false;
          const bool ok = contender && g == gstar;
          if (ok) {
            int32_t* tp = rec_tag(rec, Q, E.cand_id);
            // no-return atomicMin (agent scope) on the tag; a later holder (tag > seed here) is invalid now
            asm volatile("global_atomic_smin %0, %1, off" ::"v"(tp), "v"(seed) : "memory");
            if (E.tg != INF) {
              int32_t* dp = dead + E.tg;
              const int thief = seed + 1;
              asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dp), "v"(thief) : "memory");
            }
            vptr = tp;
            pendv = true;
          }
          PROBE(4);
          sp -= (gstar >= 0 ? gstar : E.ngv - 1) + 1;
          PROBE(5);
          if (gstar >= 0) {
#ifdef BS_PROBE
            nexp++;
#endif
            PROBE(6);
            expand(E, ok, am, gstar, cnt);
            PROBE(8);
          }
        }
        if (leave)
          break;
      }
#ifdef BS_PROBE
      if (lane == 0 && ln > 20000) {
        for (int i = 0; i < 10; i++)
          atomicAdd(&g_prof2[i], (unsigned long long)pacc[i]);
        atomicAdd(&g_prof2[10], (unsigned long long)ncalls);
        atomicAdd(&g_prof2[11], (unsigned long long)nexp);
      }
#endif
    }
    if (need_state) {  // state of the very last expansion (the plane's reported normal / centre)
      unsigned long long mm = d_am;
      while (mm) {
        const int l = __ffsll(mm) - 1;
        mm &= mm - 1;
        Sx += readlane_f64(d_mx, l);
        Sy += readlane_f64(d_my, l);
        Sz += readlane_f64(d_mz, l);
        Cx += (uint32_t)readlane_i32(d_px, l);
        Cy += (uint32_t)readlane_i32(d_py, l);
        Cz += (uint32_t)readlane_i32(d_pz, l);
      }
      plane_state_lanes(lane, Sx, Sy, Sz, Cx, Cy, Cz, (uint32_t)ln, cnx, cny, cnz, ccx, ccy, ccz);
      need_state = false;
    }
    if (have_mem && status != ST_NOMEM && !flush_list(ln))  // the whole list is in the HBM slab from here on
      status = ST_NOMEM;
    if (status == ST_DONE) {
      const bool lost = pendv && ld_i32(vptr) != seed;
      if (ballot64(lost))
        status = ST_STOLEN;
    }
    bool seed_taken = false;
    if (status == ST_STOLEN) {  // (uniform loads, said explicitly: a divergent exit of the attempt loop costs the whole kernel its scalars)
      const int t0 = __builtin_amdgcn_readfirstlane(ld_i32(rec_tag(rec, Q, seed_s)));
      seed_taken = t0 < seed && __builtin_amdgcn_readfirstlane(ld_i32(dead + (t0 < seed ? t0 : seed))) >= 0;
    }
    if (status != ST_STOLEN || seed_taken || attempt >= MAX_RETRY || ln > retry_max_list ||
        (ln > RETRY_MAX_LIST && ++long_tries > MAX_RETRY_LONG))
      break;
    for (int t0 = 1; t0 < ln; t0 += 64) {
      const int t = t0 + lane;
      if (t < ln)
        atomicCAS(rec_tag(rec, Q, ld_i32(pool.base + list.off + t)), seed, INF);
    }
    __hip_atomic_store(dead + seed, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }  // attempt loop
  if (lane == 0 && status == ST_STOLEN) {
    const int t = ld_i32(dead + seed);
    __hip_atomic_store(dead + seed, t > 0 ? -t : (t == 0 ? -0x40000000 : t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    o.steps = iters;
    o.seed = seed;
    o.status = status;
    o.keep = (status == ST_DONE && ln > a.th_count) ? 1 : 0;
    o.consistent = 0;
    o.pad = 0;
    const int dflag = ld_i32(dead + seed);
    o.thief = (dflag < 0 ? -dflag : dflag) - 1;
    o.seed_pos = seed_s;
    o.v3ok = 1;
    o.pad4 = 0;
    o.t_start = t_start;
    o.t_end = (int64_t)wall_clock64();
    o.w = w;
    // where the wave ran (debug print only): HW_ID bits [15:0] (wave, SIMD, CU, SH, SE) | XCC_ID << 16
    o.pad5 = (int32_t)((__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xffff) |
                       ((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf) << 16));
    out[w] = o;
  }
}

// The per-plane list kernels run one BLOCK of VT threads per plane: a 100 k-point
// list walked by a single wave is 1700 dependent gathers per lane (0.7 ms).
constexpr int VT = 1024;

// every accepted point must still carry this plane's claim
__global__ __launch_bounds__(VT) void validate1_kernel(PlaneOut* out, int ncand, const int32_t* __restrict__ pool,
                                                       int4* rec, int quads, const int32_t* __restrict__ dead, int64_t n,
                                                       int32_t* __restrict__ vmark, int32_t round_key, int* rejects)
{
  const int w = blockIdx.x;
  if (w >= ncand || out[w].status != ST_DONE)
    return;
  const PlaneOut o = out[w];
  // lost a point after it had finished, or an impossible list (each point at most once + the seed)
  const bool robbed = dead[o.seed] != 0 || o.list_n > n + 1;
  bool bad = robbed, bad_tag = false, bad_dup = false;
  // ... and no point may be listed twice (the claim protocol inside the launch is advisory; a plane that
  // reclaimed a point it already held would pass the tag test).  vmark[p] receives a key that is unique
  // per (round, plane): meeting one's own key again is a duplicate.  The seed's second appearance (quirk
  // Q1: it is pushed at position 0 without a label and can be accepted once more) is legal, once.
  const int32_t key = round_key + w + 1;
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += VT) {
    const int32_t p = pool[o.list_off + t];
    const bool tagbad = *rec_tag(rec, quads, p) != o.seed;
    bad_tag = bad_tag || tagbad;
    const bool dup = atomicExch(&vmark[p], key) == key;
    if (dup)
      atomicAdd(rejects, 1);
    bad_dup = bad_dup || dup;
    bad = bad || tagbad || dup;
  }
  const int b = __syncthreads_or(bad);  // (also orders the reads of out[w] above before the write below)
  const int bt = __syncthreads_or(bad_tag), bd = __syncthreads_or(bad_dup);
  if (threadIdx.x == 0) {
    if (b) {
      // which check refused the plane (bs_timings.rej_v1_*): [8] robbed after it finished, [9] a list entry without the
      // plane's claim, [10] a point listed twice
      atomicAdd(rejects + (robbed ? 8 : (bd ? 10 : 9)), 1);
      (void)bt;
      out[w].status = ST_STOLEN;
      if (rejects[1] == o.seed + 1)
        rejects[2] = 1;  // the plane the self-test forged was refused
    } else {
      out[w].pad = 1;  // insert command for plane_apply_kernel
    }
  }
}

// validate3: the plane's reported state must follow from its list.  The reference re-sums every
// member in pointIdx order (my_function.cpp:241-250): normal = S / sqrt(S.S) with S the f64 sum of the
// members' normals IN LIST ORDER from +0 (quirk Q7), centre = (int32)((uint64)(int64)(int32)C / size)
// with C the wrapping int32 coordinate sum (quirk Q3).  Both are recomputed here from the finished list
// and the records alone -- independently of the running sums the growth kernel carried -- and must
// equal the reported values bit for bit.  Together with validate1 (every entry still carries the
// plane's claim, no entry twice) this rejects the failure signature of every claim-protocol bug seen
// so far (a point held twice: duplicated entry, wrong normal / centre) instead of relying on repeated
// fuzzing to expose it.  The f64 sum is inherently sequential: wave 0 stages 64 normals per trip in
// LDS and one lane adds them in order (three independent chains); the other waves form the
// order-independent integer sum.
constexpr int V3T = 256;
constexpr int V3C = 1024;  // list entries per staged chunk
__global__ __launch_bounds__(V3T) void validate3_kernel(PlaneOut* out, int ncand, const int32_t* __restrict__ pool,
                                                        const int4* __restrict__ rec, int quads, int* rejects)
{
  __shared__ double sn[2][V3C][3];  // two staged chunks of normals (48 KB)
  __shared__ uint32_t sc[V3T / 64][3];
  const int w = blockIdx.x;
  if (w >= ncand || out[w].status != ST_DONE || out[w].pad != 1)  // pad == 1: passed validate1
    return;
  const PlaneOut o = out[w];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // The f64 sums must be taken in list order (one chain of dependent additions per coordinate): wave 0 walks
  // them.  What bounds the kernel is feeding that chain: a list entry is two dependent gathers (position,
  // then record) of ~1 us each, so the whole block stages V3C entries at a time into LDS, double-buffered --
  // the gathers of chunk c + 1 are in flight while wave 0 adds chunk c.  (One wave staging 64 entries ahead
  // spent 1.3 us per 64 entries waiting: 2.2 ms for the facade's 108 k-point plane; this form 0.5 ms.)
  // The wrapping integer sums ride along on the same gathers.
  uint32_t cx = 0, cy = 0, cz = 0;
  double rr[V3C / V3T][3];
  auto fetch = [&](int64_t c) {
#pragma unroll
    for (int u = 0; u < V3C / V3T; u++) {
      const int64_t t = c * V3C + u * V3T + tid;
      if (t < o.list_n) {
        const int4* r = rec + (int64_t)pool[o.list_off + t] * quads;
        const int4 q0 = r[0], q1 = r[1], q2 = r[2];
        cx += (uint32_t)q0.x;
        cy += (uint32_t)q0.y;
        cz += (uint32_t)q0.z;
        rr[u][0] = __hiloint2double(q1.y, q1.x);
        rr[u][1] = __hiloint2double(q1.w, q1.z);
        rr[u][2] = __hiloint2double(q2.y, q2.x);
      }
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int u = 0; u < V3C / V3T; u++) {
      sn[buf][u * V3T + tid][0] = rr[u][0];
      sn[buf][u * V3T + tid][1] = rr[u][1];
      sn[buf][u * V3T + tid][2] = rr[u][2];
    }
  };
  const int64_t chunks = (o.list_n + V3C - 1) / V3C;
  // lane c (0, 1, 2) of wave 0 carries coordinate c, so ONE LDS read + ONE v_add_f64 per list entry serve all
  // three chains (every lane runs the same code on coordinate lane % 3: no divergence, the copies are ignored)
  double acc = 0.0;
  fetch(0);
  put(0);
  __syncthreads();
  for (int64_t c = 0; c < chunks; c++) {
    const int buf = (int)(c & 1);
    if (c + 1 < chunks)
      fetch(c + 1);
    if (wv == 0) {
      const int cnt = (int)((o.list_n - c * V3C < V3C) ? o.list_n - c * V3C : V3C);
      const double* src = &sn[buf][0][lane % 3];
      int e0 = 0;
      if (cnt == V3C) {
        // batches of 16 entries, the reads of the next batch issued BEFORE the additions of the current one
        // (left to itself the compiler emits read - wait - two additions - read ...: one full LDS latency per
        // two entries, 45 cycles per entry; this form is bound by the chain of additions)
        double va[16], vb[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
          va[i] = src[i * 3];
        for (int b = 0; b < V3C / 16; b += 2) {
#pragma unroll
          for (int i = 0; i < 16; i++)
            vb[i] = src[((b + 1) * 16 + i) * 3];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 16; i++)
            acc += va[i];
          __builtin_amdgcn_sched_barrier(0);
          if (b + 2 < V3C / 16) {
#pragma unroll
            for (int i = 0; i < 16; i++)
              va[i] = src[((b + 2) * 16 + i) * 3];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 16; i++)
            acc += vb[i];
          __builtin_amdgcn_sched_barrier(0);
        }
        e0 = V3C;
      }
      for (; e0 < cnt; e0++)
        acc += src[e0 * 3];
    }
    if (c + 1 < chunks)
      put(buf ^ 1);
    __syncthreads();  // chunk c + 1 is staged, chunk c's buffer is free again
  }
  for (int off = 32; off > 0; off >>= 1) {
    cx += (uint32_t)__shfl_xor((int)cx, off);
    cy += (uint32_t)__shfl_xor((int)cy, off);
    cz += (uint32_t)__shfl_xor((int)cz, off);
  }
  if (lane == 0) {
    sc[wv][0] = cx;
    sc[wv][1] = cy;
    sc[wv][2] = cz;
  }
  const double Sx = __shfl(acc, 0), Sy = __shfl(acc, 1), Sz = __shfl(acc, 2);  // meaningful in wave 0 (tid 0 reads them)
  __syncthreads();
  if (tid == 0) {
    uint32_t Cx = 0, Cy = 0, Cz = 0;
    for (int k = 0; k < V3T / 64; k++) {
      Cx += sc[k][0];
      Cy += sc[k][1];
      Cz += sc[k][2];
    }
    bool same;
    if (o.list_n == 1) {  // a seed alone never finishes as ST_DONE; kept for completeness
      same = true;
    } else {
      const double nrm = __builtin_sqrt((Sx * Sx) + (Sy * Sy) + (Sz * Sz));
      const double nx = Sx / nrm, ny = Sy / nrm, nz = Sz / nrm;
      const CenterDiv cd = center_div_prepare((uint32_t)o.list_n);
      const int32_t ex = center_div((int32_t)Cx, cd), ey = center_div((int32_t)Cy, cd), ez = center_div((int32_t)Cz, cd);
      same = __double_as_longlong(nx) == __double_as_longlong(o.normal[0]) &&
             __double_as_longlong(ny) == __double_as_longlong(o.normal[1]) &&
             __double_as_longlong(nz) == __double_as_longlong(o.normal[2]) && ex == o.center[0] && ey == o.center[1] &&
             ez == o.center[2];
    }
    if (!same) {
      out[w].v3ok = 0;  // the host treats the plane as inconsistent: dropped from the structure, grown again
      atomicAdd(rejects, 1);
      atomicAdd(rejects + 11, 1);  // bs_timings.rej_v3_state
      if (rejects[1] == o.seed + 1)
        rejects[2] = 1;
    }
  }
}

// Self-test hook (bs_selftest_forge_next): corrupts the FIRST finished plane of a round the way the
// claim-protocol bugs of round 1 did, so that a test can watch the validation reject it.
//   mode 1: one list entry is appended once more (a point held twice);
//   mode 2: the reported normal is off by one unit in the last place.
__global__ void forge_kernel(PlaneOut* out, int ncand, int32_t* pool, int mode, int* forged)
{
  if (blockIdx.x != 0 || threadIdx.x != 0)
    return;
  for (int w = 0; w < ncand; w++) {
    const int64_t ln = out[w].list_n;
    if (out[w].status != ST_DONE || ln < 64 || (ln & (ln - 1)) == 0)  // (a power of two may fill its slab exactly)
      continue;
    if (mode == 1) {
      pool[out[w].list_off + ln] = pool[out[w].list_off + 20];  // entry 20 once more, at the end
      out[w].list_n = ln + 1;
    } else
      out[w].normal[2] = __longlong_as_double(__double_as_longlong(out[w].normal[2]) ^ 1ll);
    *forged = out[w].seed + 1;
    return;
  }
}

// A finished plane enters the owner structure: its seed stops being an orphan
// maker and its kept points get the plane as base owner.  drop undoes it.
// (PlaneOut.pad carries the host's per-plane command: 1 = insert, 2 = drop.)
__global__ __launch_bounds__(VT) void plane_apply_kernel(const PlaneOut* __restrict__ arr, int cnt,
                                                         const int32_t* __restrict__ pool, int32_t* base,
                                                         uint8_t* ps, uint8_t* dirty, uint8_t* bdirty, int32_t* omega,
                                                         const uint32_t* __restrict__ occ, int4* rec, int quads)
{
  const int w = blockIdx.x;
  if (w >= cnt)
    return;
  const PlaneOut o = arr[w];
  if (o.status != ST_DONE || o.pad == 0)
    return;
  const bool ins = o.pad == 1;
  if (threadIdx.x == 0) {
    ps[o.seed_pos] = ins ? 1 : 0;
    dirty[o.seed_pos] = 1;
    bdirty[o.seed_pos >> 8] = 1;
  }
  if (!o.keep)
    return;
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += VT) {
    const int32_t p = pool[o.list_off + t];
    if (ins) {
      // Lowering base[p] to the plane's seed lowers the owner to min(owner, seed) EXACTLY -- no look at the
      // reverse list is needed.  Only a point that currently occurs as an orphan maker has to be re-evaluated
      // (it stops occurring and its neighbours must hear about it): ~1 in 10 points of a fresh plane instead of
      // all of them.
      atomicMin(&base[p], o.seed);
      const int32_t old = atomicMin(&omega[p], o.seed);
      if (old > o.seed)
        reinterpret_cast<int32_t*>(rec + (int64_t)p * quads)[3] = o.seed;
      if ((occ[p >> 5] >> (p & 31)) & 1u) {
        dirty[p] = 1;
        bdirty[p >> 8] = 1;
      }
    } else {
      atomicCAS(&base[p], o.seed, INF);  // the owner may rise: full re-evaluation
      dirty[p] = 1;
      bdirty[p >> 8] = 1;
    }
  }
}

// End of a round: every claim goes back to "free".  A plane's claims are its list
// (any status: finished, stolen, rolled back) plus -- for a seed that failed at depth
// 0 -- neighbours of the seed that were claimed in the very step that failed and
// never reached the list.  Replaces a full pass over all records per round.
__global__ __launch_bounds__(VT) void reset_tags_kernel(const PlaneOut* __restrict__ out, int ncand,
                                                        const int32_t* __restrict__ pool, int4* rec, int quads, int K)
{
  const int w = blockIdx.x;
  if (w >= ncand)
    return;
  const PlaneOut o = out[w];
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += VT) {
    int32_t* tg = rec_tag(rec, quads, pool[o.list_off + t]);
    if (*tg == o.seed)
      *tg = INF;
  }
  if (threadIdx.x >= 1 && threadIdx.x < K) {
    int32_t* tg = rec_tag(rec, quads, reinterpret_cast<const int32_t*>(rec + (int64_t)o.seed_pos * quads + 4)[threadIdx.x]);
    if (*tg == o.seed)
      *tg = INF;
  }
}

// BS_VERIFY=1: do the incrementally maintained records equal a full refresh?
__global__ void verify_records_kernel(const int32_t* __restrict__ omega, int4* rec, int quads, int64_t n, int* nbad)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  if (reinterpret_cast<int32_t*>(rec + i * quads)[3] != omega[i] || *rec_tag(rec, quads, i) != INF)
    atomicAdd(nbad, 1);
}

__global__ __launch_bounds__(VT) void validate2_kernel(PlaneOut* out, int ncand, const int32_t* __restrict__ pool,
                                                       const int32_t* __restrict__ omega, const int4* __restrict__ rec,
                                                       int quads, int K)
{
  const int w = blockIdx.x;
  if (w >= ncand || out[w].status != ST_DONE)
    return;
  const PlaneOut o = out[w];
  const int32_t s = o.seed;
  bool bad1 = false, bad2 = false, bad3 = false;
  if (threadIdx.x == 0)
    bad1 = omega[o.seed_pos] < s;  // (1) the seed is still free at its time ...
  if (threadIdx.x >= 1 && threadIdx.x < K)  // ... and so are its K-1 neighbours
    bad1 = omega[reinterpret_cast<const int32_t*>(rec + (int64_t)o.seed_pos * quads + 4)[threadIdx.x]] < s;
  for (int64_t t = 1 + threadIdx.x; t < o.list_n; t += VT)  // (2) accepted points were free
    bad2 = bad2 || omega[pool[o.list_off + t]] < s;
  for (int64_t t = threadIdx.x; t < o.log_n; t += VT)  // (3) assumed-taken points are taken
    bad3 = bad3 || !(omega[pool[o.log_off + t]] < s);
  const int b = __syncthreads_or((bad1 ? 1 : 0) | (bad2 ? 2 : 0) | (bad3 ? 4 : 0));
  const int b2 = __syncthreads_or(bad2), b3 = __syncthreads_or(bad3);
  if (threadIdx.x == 0) {
    out[w].consistent = b ? 0 : 1;
    out[w].pad4 = (b && !b2 && !b3 ? 1 : 0) | (b2 ? 2 : 0) | (b3 ? 4 : 0);  // diagnostics (BS_DEBUG): which test failed
  }
}

// ---- audit (bs_set_audit): replayed attempt vs. what was committed ---------------------------
// out[w] is the attempt of seed out[w].seed grown again against the FINAL owners (no speculation).
// A committed seed must reproduce its list entry by entry (the replay holds positions, the committed
// list original indices), its normal and its centre bit for bit; any other attempt must be one the
// reference rolls back (list_n <= th_count).  stats[0] = mismatches, stats[1] = committed planes met.
__global__ __launch_bounds__(VT) void audit_compare_kernel(const PlaneOut* __restrict__ out, int na,
                                                           const int32_t* __restrict__ pool, const int32_t* __restrict__ prio,
                                                           const int32_t* __restrict__ seeds, const PlaneRec* __restrict__ planes,
                                                           int np, const int32_t* __restrict__ lists, int64_t th_count, int* stats,
                                                           const int32_t* __restrict__ dead, int32_t* __restrict__ retry)
{
  const int w = blockIdx.x;
  if (w >= na)
    return;
  const PlaneOut o = out[w];
  if (retry) {
    // batched replay of attempts the reference rolls back: they may meet each other.  Whoever was robbed, or treated a
    // point as held by another attempt of the batch (logged), did NOT see the sequential state: it is replayed again
    // in the next batch instead of being judged (the lowest seed of a batch is never interfered with).
    const bool interfered = o.status != ST_DONE || o.log_n != 0 || dead[o.seed] != 0;
    if (threadIdx.x == 0)
      retry[w] = interfered ? 1 : 0;
    if (interfered)
      return;
  }
  int lo = 0, hi = np;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seeds[mid] < o.seed)
      lo = mid + 1;
    else
      hi = mid;
  }
  const bool committed = lo < np && seeds[lo] == o.seed;
  bool bad = o.status != ST_DONE;
  if (!bad && committed) {
    const PlaneRec r = planes[lo];
    bad = r.list_n != o.list_n || r.center[0] != o.center[0] || r.center[1] != o.center[1] || r.center[2] != o.center[2] ||
          __double_as_longlong(r.normal[0]) != __double_as_longlong(o.normal[0]) ||
          __double_as_longlong(r.normal[1]) != __double_as_longlong(o.normal[1]) ||
          __double_as_longlong(r.normal[2]) != __double_as_longlong(o.normal[2]);
    if (!bad)
      for (int64_t t = threadIdx.x; t < o.list_n; t += VT)
        bad = bad || lists[r.list_off + t] != prio[pool[o.list_off + t]];
  } else if (!bad) {
    bad = o.list_n > th_count;
  }
  const int b = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    if (b)
      atomicAdd(stats, 1);
    if (committed)
      atomicAdd(stats + 1, 1);
  }
}

struct CopyDesc {
  const int32_t* src;
  int32_t* dst;
  int64_t cnt;
  const int32_t* map;  // committed lists leave position space: dst = map[src] (null: plain copy)
};

__global__ void copy_lists_kernel(const CopyDesc* __restrict__ d, int nd)
{
  const int w = blockIdx.x;
  if (w >= nd)
    return;
  const CopyDesc c = d[w];
  for (int64_t i = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; i < c.cnt; i += (int64_t)gridDim.y * blockDim.x)
    c.dst[i] = c.map ? c.map[c.src[i]] : c.src[i];
}

__global__ void label_kernel(const int32_t* __restrict__ owner, int64_t n, const int32_t* __restrict__ seeds,
                             int np, const int32_t* __restrict__ prio, int32_t* __restrict__ plane_idx)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t o = owner[i];
  const int64_t dst = prio[i];  // i is a position; the caller's label array is in original order
  if (o == INF) {
    plane_idx[dst] = -1;
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
  plane_idx[dst] = 1 + lo;
}

// ---- dispatch order of a big round ------------------------------------------------------------
// A round lasts as long as its longest plane, and a workgroup starts when a slot is free: with 158 k attempts
// queued in seed order the five longest planes of the 50 M cloud's first round started 13-16 ms late, behind
// attempts that sit INSIDE a plane another, lower seed is growing (they grow until its front reaches them:
// 115 us of slot time on average, 45 s of wave time in total over 2 560 slots).  The seed of a plane is the
// lowest candidate inside it, so the lowest candidate of a tile of positions (Morton order: a compact piece of
// space) is probably a real seed: those go first -- tiles of 2^16 positions, then of 2^12 -- and the rest
// follows in seed order.  Only the dispatch order changes; the attempts keep their numbers (w = rank by seed).
constexpr int TILE1 = 12, TILE2 = 16;
__global__ void tile_min_kernel(const unsigned long long* __restrict__ cand, int ncand, unsigned long long* __restrict__ tmin1,
                                unsigned long long* __restrict__ tmin2)
{
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= ncand)
    return;
  const unsigned long long c = cand[w];
  const uint32_t pos = (uint32_t)c;
  atomicMin(&tmin1[pos >> TILE1], c);
  atomicMin(&tmin2[pos >> TILE2], c);
}

__global__ void dispatch_key_kernel(const unsigned long long* __restrict__ cand, int ncand, const unsigned long long* __restrict__ tmin1,
                                    const unsigned long long* __restrict__ tmin2, uint32_t* __restrict__ keys)
{
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= ncand)
    return;
  const unsigned long long c = cand[w];
  const uint32_t pos = (uint32_t)c;
  const uint32_t lvl = tmin2[pos >> TILE2] == c ? 0u : (tmin1[pos >> TILE1] == c ? 1u : 2u);
  keys[w] = lvl * (uint32_t)MAX_WAVES + (uint32_t)w;
}

__global__ void reset_dead_kernel(const unsigned long long* __restrict__ cand, int ncand, int32_t* __restrict__ dead)
{
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < ncand)
    dead[(int32_t)(cand[w] >> 32)] = 0;
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = v;
}

inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

struct ToI64 {
  __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

}  // namespace

int launch_region_grow_spec(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals, const int32_t* d_neigh,
                            int64_t n, const bs_params& p, int32_t* d_plane_idx)
{
  hipStream_t st = ctx->stream;
  ctx->rg_valid = false;
  ctx->rg_omega = nullptr;
  ctx->rg_prio = nullptr;
  ctx->rg_seeds = nullptr;
  ctx->rg_nplanes = 0;
  const int K = p.k;
  const int64_t list_cap = 2 * n + 64;
  const int64_t planes_cap = n / std::max(1, p.th_point_count) + 64;
  // round pool: lists + stacks + logs of every concurrent attempt
  // concurrent plane attempts per round: more waves let late-index planes start in earlier rounds
  int retry_max_list = RETRY_MAX_LIST;
  bool retry_env = false;
  if (const char* e = getenv("BS_RETRY_MAX_LIST")) {
    retry_max_list = atoi(e);
    retry_env = true;
  }
  int retry_big_round = 1024;
  if (const char* e = getenv("BS_RETRY_BIG_ROUND"))
    retry_big_round = atoi(e);
  // Plane attempts per round: ALL candidates.  A round is a barrier bounded by its longest plane, and the big
  // planes of a large scene are independent of each other -- but their seeds are spread over the whole index
  // range: with the lowest 32 768 candidates per round (round 1) the 50 M cloud grew its 136 k-, 131 k- and
  // 122 k-entry planes one round AFTER the 80 k-entry one although nothing connected them (39 k + 68 k + 35 k
  // sequential steps; a cap of 65 536 was worse still: 68 k + 66 k + 35 k).  The ~90 % of the attempts that fail
  // at depth 0 cost one step and 1.5 KB of pool each.  (Bounded at 2^18: with 655 k attempts on the 200 M scene the
  // round pool outgrew 16 GB and the FIRST call on a fresh context took 3 s longer; 2^18 keeps the pool at 12 GB.)
  int max_waves = MAX_WAVES;
  if (const char* e = getenv("BS_MAX_WAVES"))
    max_waves = atoi(e);
  const int wave_cap = (int)std::max<int64_t>(1, std::min<int64_t>(MAX_WAVES, n / 8 + 64));  // sizes the per-attempt arrays
  max_waves = (int)std::max<int64_t>(1, std::min<int64_t>(max_waves, wave_cap));
  const unsigned long long pool_cap = (unsigned long long)std::max<int64_t>(
      std::min<int64_t>(64 * n, (int64_t)3 << 30), 8 * n + (int64_t)(2 * 256 + 32 * 32) * 2 * max_waves);
  BS_HIP(ctx, ctx->rg_list.reserve(sizeof(int32_t) * list_cap));
  BS_HIP(ctx, ctx->rg_planes.reserve(sizeof(PlaneRec) * planes_cap));
  BS_HIP(ctx, ctx->rg_stats.reserve(sizeof(GrowStats)));
  // aux layout (int32 units): misc[1024] | hmask | omega | base | dead | prio | rpos(n+64) | vmark | seeds |
  //                            (u8) flags | ps | occ | dirty0 | dirty1 | PlaneOut[MAX_WAVES + MAX_PENDING] | CopyDesc[]
  const size_t n_i32 = (size_t)(7 * n + planes_cap + 1024 + 128);
  const size_t nb256 = (size_t)((n + 255) / 256);
  const size_t aux_bytes = sizeof(int32_t) * n_i32 + (size_t)5 * n + 3 * nb256 + 8192 +
                           sizeof(PlaneOut) * ((size_t)wave_cap + MAX_PENDING) +
                           sizeof(CopyDesc) * ((size_t)wave_cap + MAX_PENDING);
  BS_HIP(ctx, ctx->rg_aux.reserve(aux_bytes));
  BS_HIP(ctx, ctx->rg_stack.reserve(sizeof(int32_t) * pool_cap));
  // reverse lists: <= n (k - 1) entries with 64-bit offsets (200 M points at k = 16 are 3.0e9 entries)
  BS_HIP(ctx, ctx->rg_radj.reserve(sizeof(int32_t) * (size_t)(n * (K - 1) + 16)));
  BS_HIP(ctx, ctx->rg_roff.reserve(sizeof(int64_t) * (size_t)(n + 2)));
  int64_t* roff = ctx->rg_roff.as<int64_t>();
  int32_t* aux = ctx->rg_aux.as<int32_t>();
  int32_t* d_misc = aux;  // [0]=any [1]=ncand [2]=min_idx ; [16..17] pool top (u64)
  uint32_t* hmask = (uint32_t*)(aux + 1024);
  int32_t* omega = aux + 1024 + n;
  int32_t* base = aux + 1024 + 2 * n;
  int32_t* dead = aux + 1024 + 3 * n;
  int32_t* prio = aux + 1024 + 4 * n;    // original index of every position (n entries)
  int32_t* rpos = aux + 1024 + 5 * n;    // n + 1 (+ pad): reverse-list counts / fill cursors, later the candidate scratch
  int32_t* vmark = aux + 1024 + 6 * n + 64;    // duplicate detection of validate1 (n entries)
  int32_t* d_seeds = aux + 1024 + 7 * n + 64;  // committed seeds (planes_cap)
  uint8_t* flags = (uint8_t*)(aux + n_i32);
  uint8_t* ps = flags + n;
  uint32_t* occ = (uint32_t*)(((uintptr_t)(ps + n) + 15) & ~(uintptr_t)15);  // bitmap, 16-byte aligned: (n + 31) / 32 words
  uint8_t* dirty0 = (uint8_t*)occ + ((n + 31) / 32) * 4 + 16;
  uint8_t* dirty1 = dirty0 + n;
  uint8_t* bdirty0 = dirty1 + n;  // one flag per 256 points
  uint8_t* bdirty1 = bdirty0 + nb256;
  uint8_t* bcand = bdirty1 + nb256;  // one flag per 256 positions: can the group still hold a plane seed?
  PlaneOut* d_out = (PlaneOut*)(((uintptr_t)(bcand + nb256) + 255) & ~(uintptr_t)255);
  PlaneOut* d_pend = d_out + wave_cap;
  CopyDesc* d_copy = (CopyDesc*)(d_pend + MAX_PENDING);
  int32_t* radj = ctx->rg_radj.as<int32_t>();
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
  a.pad = 0;

  const int KC = K <= 16 ? 16 : 32;
  const int quads = 4 + KC / 4;
  BS_HIP(ctx, ctx->rg_rec.reserve(sizeof(int4) * (size_t)quads * n));
  int4* rec = ctx->rg_rec.as<int4>();
  // scratch: the 64-bit fill cursors of the reverse lists during setup, then the candidate lists
  // (unsorted | sorted, one (original index << 32 | position) key each)
  BS_HIP(ctx, ctx->rg_geo.reserve(sizeof(unsigned long long) * 2 * (size_t)(n + 64)));
  unsigned long long* cand_raw = ctx->rg_geo.as<unsigned long long>();
  BS_HIP(ctx, ctx->rg_gs.reserve(sizeof(int4) * 3 * (size_t)(n + 1)));  // compact geometry by position (setup only)
  int4* gs = ctx->rg_gs.as<int4>();
  unsigned long long* d_cand = cand_raw + n + 64;
  // dispatch order of big rounds: tile minima + sort keys
  const size_t nt1 = (size_t)(n >> TILE1) + 2, nt2 = (size_t)(n >> TILE2) + 2;
  BS_HIP(ctx, ctx->rg_disp.reserve(sizeof(unsigned long long) * (nt1 + nt2) + sizeof(uint32_t) * 2 * (size_t)wave_cap + 64 +
                                    sizeof(PlaneOut) * (size_t)wave_cap));
  unsigned long long* tmin1 = ctx->rg_disp.as<unsigned long long>();
  unsigned long long* tmin2 = tmin1 + nt1;
  uint32_t* dkeys_in = reinterpret_cast<uint32_t*>(tmin2 + nt2);
  uint32_t* dkeys_out = dkeys_in + wave_cap;
  PlaneOut* d_outc = reinterpret_cast<PlaneOut*>(((uintptr_t)(dkeys_out + wave_cap) + 15) & ~(uintptr_t)15);  // finished attempts of a big round, compacted
  const bool dispatch_order = getenv("BS_NO_DISPATCH_ORDER") == nullptr;  // developer A/B switch
  // positions = the search grid's cell-sorted (Morton) order of THIS cloud when it is cached on the context
  const int32_t* order = (ctx->order_n == n && ctx->order_xyz == d_xyz) ? ctx->vals_out.as<int32_t>() : nullptr;
  int32_t* pos = nullptr;
  const int32_t* npos = (order && ctx->npos_neigh == d_neigh && ctx->npos_k == K) ? ctx->seg_npos.as<int32_t>() : nullptr;
  (void)hipEventRecord(ctx->ev[8], st);
  if (!npos) {
    // Foreign input (bs_region_grow[_dev], the component-sharded stage 3): no search grid of THIS cloud's current
    // contents is at hand (a cached order is keyed by pointer only).  The grower builds the Morton order itself --
    // bounding box, keys, one radix sort -- instead of falling back to the identity, where every neighbour of every
    // pass is a random HBM access (setup 134 ms instead of 31 at 50 M, HBM-latency steps).
    static const bool identity = getenv("BS_FOREIGN_ORDER_IDENTITY") != nullptr;  // developer A/B switch
    order = nullptr;
    if (!identity) {
      const int orc = build_spatial_order(ctx, d_xyz, n);
      if (orc != BS_OK)
        return orc;
      order = ctx->vals_out.as<int32_t>();
    }
  }
  // (npos only exists for a cloud the fused pipeline just searched WITHOUT global-index indirection: spts[s].w is
  // the original index then, and the position-ordered normals belong to exactly this normals buffer)
  const int4* spts_in = npos ? ctx->spts.as<int4>() : nullptr;
  const double* pnorm_in = (npos && ctx->npos_normals == d_normals) ? pnorm_of(ctx->seg_npos.as<int32_t>(), n, K) : nullptr;
  if (order && !npos) {
    pos = vmark;  // (free until the validation marks are cleared below)
    invert_order_kernel<<<nblk(n, 256), 256, 0, st>>>(order, n, pos);
  }
  if (KC == 16)
    build_records_kernel<16><<<nblk(n, 256), 256, 0, st>>>(a, order, pos, npos, rec, prio, gs, spts_in, pnorm_in);
  else
    build_records_kernel<32><<<nblk(n, 256), 256, 0, st>>>(a, order, pos, npos, rec, prio, gs, spts_in, pnorm_in);
  // static masks + reverse-list counts in one pass, offsets by a 64-bit exclusive scan over n + 1
  // entries (roff[n] = total), then the fill
  BS_HIP(ctx, hipMemsetAsync(rpos, 0, sizeof(int32_t) * (n + 1), st));
  uint32_t* lmask = reinterpret_cast<uint32_t*>(vmark);  // (free until the validation marks are cleared after the fill)
  const int32_t* rows = npos ? npos : reinterpret_cast<const int32_t*>(rec) + 16;
  const int row_stride = npos ? K : quads * 4;
  static_mask_kernel<<<xcd_grid(nblk(n, 256)), 256, 0, st>>>(a, rows, row_stride, gs, hmask, rpos, lmask);
  {
    hipcub::TransformInputIterator<int64_t, ToI64, const int32_t*> in(rpos, ToI64());
    size_t tb = 0;
    BS_HIP(ctx, hipcub::DeviceScan::ExclusiveScan(nullptr, tb, in, roff, hipcub::Sum(), (int64_t)0, (int)(n + 1), st));
    BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
    BS_HIP(ctx, hipcub::DeviceScan::ExclusiveScan(ctx->cub_tmp.p, tb, in, roff, hipcub::Sum(), (int64_t)0, (int)(n + 1), st));
  }
  unsigned long long* rcur = cand_raw;
  BS_HIP(ctx, hipMemcpyAsync(rcur, roff, sizeof(int64_t) * n, hipMemcpyDeviceToDevice, st));
  rev_fill_kernel<<<xcd_grid(nblk(n, 256)), 256, 0, st>>>(hmask, rows, row_stride, n, rcur, radj, lmask);
  // initial state: no plane; the first owner fixed point is computed by decided states (see
  // decide_pass_kernel), after which nothing is dirty
  fill_i32_kernel<<<nblk(n, 256), 256, 0, st>>>(base, n, INF);
  seed_flags_kernel<<<nblk(n, 256), 256, 0, st>>>(hmask, n, K, ps);
  BS_HIP(ctx, hipMemsetAsync(vmark, 0, sizeof(int32_t) * n, st));
  BS_HIP(ctx, hipMemsetAsync(occ, 0, sizeof(uint32_t) * (size_t)((n + 31) / 32 + 2), st));
  int64_t passes = 0;
  // 256-point groups per workgroup of a pass: enough workgroups to fill the chip in the heavy first
  // passes, few enough that an (almost) idle pass costs microseconds
  const int pull_sub = (int)std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)nb256 / 4096));
  // the candidate scan does real work in most groups until the last rounds: fewer groups per workgroup keep it parallel
  const int cand_sub = std::max(1, pull_sub / 8);
  {
    uint8_t* state = dirty1;   // scratch until the dirty flags are cleared below
    uint8_t* bund = bdirty1;
    decide_init_kernel<<<nblk(n, 256), 256, 0, st>>>(hmask, n, state, bund);
    for (int it = 0;; it++) {
      BS_HIP(ctx, hipMemsetAsync(d_misc, 0, sizeof(int), st));
      decide_pass_kernel<<<(int)((nb256 + pull_sub - 1) / pull_sub), 256, 0, st>>>(n, pull_sub, prio, roff, radj, state, bund,
                                                                                  d_misc);
      int any = 0;
      BS_HIP(ctx, hipMemcpyAsync(&any, d_misc, sizeof any, hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      passes++;
      if (!any)
        break;
      if (it > 4 * 1000 * 1000)
        return fail(ctx, BS_ERR_INTERNAL, "orphan fixed point (decided states) did not converge");
    }
    decide_finish_kernel<<<nblk(n, 256), 256, 0, st>>>(n, prio, roff, radj, state, omega, occ, rec, quads, rpos);
  }
  BS_HIP(ctx, hipMemsetAsync(dirty0, 0, n, st));
  BS_HIP(ctx, hipMemsetAsync(dirty1, 0, n, st));
  BS_HIP(ctx, hipMemsetAsync(bdirty0, 0, nb256, st));
  BS_HIP(ctx, hipMemsetAsync(bdirty1, 0, nb256, st));
  BS_HIP(ctx, hipMemsetAsync(bcand, 1, nb256, st));
  BS_HIP(ctx, hipMemsetAsync(dead, 0, sizeof(int32_t) * n, st));
  uint8_t* dcur = dirty0;
  uint8_t* dnext = dirty1;
  uint8_t* bcur = bdirty0;
  uint8_t* bnext = bdirty1;
  // per-round results read by the host: page-locked (158 k attempts in the first round of the 50 M cloud are
  // 18 MB each way)
  BS_HIP(ctx, ctx->rg_hout.reserve(sizeof(PlaneOut) * ((size_t)wave_cap + MAX_PENDING) + 256));
  PlaneOut* const h_out = ctx->rg_hout.as<PlaneOut>();
  PlaneOut* const h_pend = h_out + wave_cap;
  int32_t* const h_flags = reinterpret_cast<int32_t*>(h_pend + MAX_PENDING);  // small scalars copied back per pass / round
  // The passes are launched in groups of PULL_GROUP with ONE host round trip per group: a pass that finds
  // nothing dirty costs microseconds, a round trip ~50 us, and a plane insertion settles in 8-25 passes.
  // Every pass reports its flips in its own word; the structure is a fixed point when the LAST pass of a
  // group flipped nothing (the passes after the settling one find no dirty point and do nothing).
  // (16 per group: a settling takes 8-26 passes, so ONE round trip usually decides it; the passes launched after
  // the settling one find nothing dirty: 16 x 2-3 us against 45 us per extra round trip with groups of four)
  constexpr int PULL_GROUP = 16;
  int32_t* const d_flip = d_misc + 32;
  int32_t* const h_flip = h_flags + 32;
  auto propagate = [&]() -> int {
    for (int it = 0; it < 1000000; it++) {
      BS_HIP(ctx, hipMemsetAsync(d_flip, 0, sizeof(int32_t) * PULL_GROUP, st));
      for (int g = 0; g < PULL_GROUP; g++) {
        pull_pass_kernel<<<(int)((nb256 + pull_sub - 1) / pull_sub), 256, 0, st>>>(n, K, pull_sub, hmask, prio, ps, base, roff, radj,
                                                                                  omega, occ, dcur, dnext, bcur, bnext, rec, quads,
                                                                                  d_flip + g, rpos /* = minr after the setup */,
                                                                                  g ? d_flip + g - 1 : nullptr);
        passes++;
        std::swap(dcur, dnext);  // dcur now holds the newly dirtied points (the old dcur was cleared by the pass)
        std::swap(bcur, bnext);
      }
      BS_HIP(ctx, hipMemcpyAsync(h_flip, d_flip, sizeof(int32_t) * PULL_GROUP, hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      if (!h_flip[PULL_GROUP - 1]) {
        if (getenv("BS_VERIFY")) {
          BS_HIP(ctx, hipMemsetAsync(d_misc + 3, 0, sizeof(int), st));
          verify_fixpoint_kernel<<<nblk(n, 256), 256, 0, st>>>(n, hmask, prio, ps, base, roff, radj, omega, occ, d_misc + 3);
          int nb = 0;
          BS_HIP(ctx, hipMemcpyAsync(&nb, d_misc + 3, sizeof nb, hipMemcpyDeviceToHost, st));
          BS_HIP(ctx, hipStreamSynchronize(st));
          if (nb)
            fprintf(stderr, "[bs] VERIFY: owner structure is not a fixed point at %d points (pass %ld)\n", nb, (long)passes);
        }
        return BS_OK;
      }
    }
    return fail(ctx, BS_ERR_INTERNAL, "orphan fixed point did not converge");
  };
  int rc = propagate();
  if (rc != BS_OK)
    return rc;
  (void)hipEventRecord(ctx->ev[9], st);

  // persistent store for planes that finished consistently but cannot be
  // finalised yet (an earlier attempt is still open): kept across rounds and
  // re-validated instead of being re-grown
  const int64_t pstore_cap = 6 * n + 65536;
  BS_HIP(ctx, ctx->rg_pstore.reserve(sizeof(int32_t) * pstore_cap));
  int32_t* pstore = ctx->rg_pstore.as<int32_t>();
  int64_t pstore_top = 0;
  std::vector<PlaneOut> pending;

  std::vector<PlaneRec> recs;
  std::vector<int32_t> seeds;
  std::vector<CopyDesc> copies;
  int64_t list_used = 0, largest = 0, attempts = 0, rounds = 0, grow_launches = 0;
  double grow_ms = 0.0;
  bool timed_round = false;
  const bool force_full_refresh = getenv("BS_FULL_REFRESH") != nullptr;  // debugging aid: the pre-incremental behaviour
  bool full_refresh = force_full_refresh;
  int32_t F = 0;
  auto commit_plane = [&](const PlaneOut& o, const int32_t* src_pool) -> int {
    attempts++;
    if (!o.keep)
      return BS_OK;  // rolled back: no trace
    if (list_used + o.list_n > list_cap || (int64_t)recs.size() >= planes_cap)
      return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): list pool overflow");
    copies.push_back({src_pool + o.list_off, ctx->rg_list.as<int32_t>() + list_used, o.list_n, prio});
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
    return BS_OK;
  };
  auto flush_copies = [&]() -> int {
    for (size_t off = 0; off < copies.size(); off += (size_t)wave_cap + MAX_PENDING) {
      const int nd = (int)std::min<size_t>(copies.size() - off, (size_t)wave_cap + MAX_PENDING);
      BS_HIP(ctx, hipMemcpyAsync(d_copy, copies.data() + off, sizeof(CopyDesc) * nd, hipMemcpyHostToDevice, st));
      copy_lists_kernel<<<dim3(nd, 8), 256, 0, st>>>(d_copy, nd);
      BS_HIP(ctx, hipStreamSynchronize(st));  // descriptors are reused
    }
    copies.clear();
    return BS_OK;
  };
  bool cand_listed = false;
  int32_t ncand_all = 0;
  bool v3_pending = false;
  // validate3 runs on the side stream beside the owner passes: whatever way this function is left (a failed HIP call,
  // the watchdog, a refused plane), nothing of it may still be in flight on the buffers the next call reuses
  struct SideGuard {
    bs_ctx* c;
    bool* pending;
    ~SideGuard()
    {
      if (*pending && c->side)
        (void)hipStreamSynchronize(c->side);
    }
  } side_guard{ctx, &v3_pending};
  bool dead_dirty = false;
  if (!ctx->side) {
    BS_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    BS_HIP(ctx, hipEventCreateWithFlags(&ctx->sev[0], hipEventDisableTiming));
    BS_HIP(ctx, hipEventCreateWithFlags(&ctx->sev[1], hipEventDisableTiming));
  }
  int32_t rejects_seen = 0;
  int refused_rounds = 0;
  int64_t incons[3] = {0, 0, 0};
  int forge_mode = ctx->forge_mode;
  ctx->forge_mode = 0;
  const int lds_pad = getenv("BS_GROW_LDS_PAD") ? atoi(getenv("BS_GROW_LDS_PAD")) : 0;  // experiment: unused dynamic LDS lowers the occupancy
  const bool dbg = getenv("BS_DEBUG") != nullptr;  // (not once per attempt: 158 k of them in a first round)
  const bool do_validate3 = getenv("BS_NO_VALIDATE3") == nullptr;  // developer A/B switch
  // Step engine.  grow_spec2_kernel (hot loop + complete step) is ~8 % faster per step on a chain of steps served
  // from L2 / Infinity Cache (facade 1 M: 136.7 vs 147.8 ms of growth kernels) but needs more registers: with k > 16
  // (255 VGPRs, one wave per SIMD) and in rounds with tens of thousands of attempts, where the throughput of the
  // many short attempts counts and the long chains wait for HBM anyway (urban 10 M k=32: 128 vs 112 ms, urban 50 M:
  // 199.5 vs 184.8 ms), the first engine wins.  So: second engine for rounds of few attempts at k <= 16.
  // BS_GROW_V2=0 / 1 forces one of them (A/B runs, tests).
  const int grow_force = getenv("BS_GROW_V2") ? atoi(getenv("BS_GROW_V2")) : -1;
  BS_HIP(ctx, hipMemsetAsync(d_misc + 4, 0, 12 * sizeof(int), st));  // [4] refused planes, [5] forged seed + 1, [6] forged one refused, [12..15] refusals by check
  for (;;) {
    rounds++;
    a.F = F;
    const int npend = (int)pending.size();
    // lowest new plane-attempt candidates under the current owners (pending
    // planes are in the structure, their seeds are not candidates)
    // (one pass: the sparse candidates are appended unordered into rpos -- free after the reverse
    // lists were built -- and the few entries are sorted; a flag array + stream compaction over
    // all n points cost 0.7 ms per round at 50 M)
    // (the list is usually left over from the end of the previous round: the pass that looks for the
    // lowest NEW candidate sees exactly the owners this round starts from unless planes were dropped)
    if (!cand_listed) {
      BS_HIP(ctx, hipMemsetAsync(d_misc + 1, 0, sizeof(int32_t), st));
      cand_flag_kernel<<<(int)((nb256 + cand_sub - 1) / cand_sub), 256, 0, st>>>(rec, quads, K, n, F, prio, ps, 0, omega, nullptr, cand_raw, d_misc + 1, bcand, cand_sub);
      BS_HIP(ctx, hipMemcpyAsync(h_flags + 11, d_misc + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      ncand_all = h_flags[11];
    }
    cand_listed = false;
    if (ncand_all == 0 && npend == 0)
      break;  // no plane attempt left: omega is the final owner array
    if (ncand_all > 0) {
      size_t tb = 0;
      BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, cand_raw, d_cand, ncand_all, 0, 64, st));
      BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
      BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(ctx->cub_tmp.p, tb, cand_raw, d_cand, ncand_all, 0, 64, st));
    }
    const int ncand = std::min<int>(ncand_all, max_waves);
    if (npend)
      BS_HIP(ctx, hipMemcpyAsync(d_pend, pending.data(), sizeof(PlaneOut) * npend, hipMemcpyHostToDevice, st));
    if (ncand > 0) {
      // grow them concurrently against the current owners
      // Records are maintained incrementally: pull_pass_kernel writes every owner change
      // into the record and reset_tags_kernel frees the claims of the previous round.  A
      // full pass is only needed after a pool exhaustion (claims of a plane that ran out
      // of memory between claiming and listing are not recorded anywhere).
      if (full_refresh) {
        refresh_records_kernel<<<nblk(n, 256), 256, 0, st>>>(omega, rec, quads, n);
        full_refresh = force_full_refresh;
      } else if (getenv("BS_VERIFY")) {
        BS_HIP(ctx, hipMemsetAsync(d_misc + 3, 0, sizeof(int), st));
        verify_records_kernel<<<nblk(n, 256), 256, 0, st>>>(omega, rec, quads, n, d_misc + 3);
        int nb = 0;
        BS_HIP(ctx, hipMemcpyAsync(&nb, d_misc + 3, sizeof nb, hipMemcpyDeviceToHost, st));
        BS_HIP(ctx, hipStreamSynchronize(st));
        if (nb)
          fprintf(stderr, "[bs] VERIFY: %d records differ from a full refresh (round %ld)\n", nb, (long)rounds);
      }
      if (dead_dirty) {  // after a pool exhaustion stale tags can point at seeds outside the round's candidate list
        BS_HIP(ctx, hipMemsetAsync(dead, 0, sizeof(int32_t) * n, st));
        dead_dirty = false;
      }
      BS_HIP(ctx, hipMemsetAsync(d_pool_top, 0, sizeof(unsigned long long), st));
      const uint32_t* d_order = nullptr;
      if (dispatch_order && ncand >= 4096) {  // see dispatch_key_kernel
        BS_HIP(ctx, hipMemsetAsync(tmin1, 0xff, sizeof(unsigned long long) * (nt1 + nt2), st));
        tile_min_kernel<<<nblk(ncand, 256), 256, 0, st>>>(d_cand, ncand, tmin1, tmin2);
        dispatch_key_kernel<<<nblk(ncand, 256), 256, 0, st>>>(d_cand, ncand, tmin1, tmin2, dkeys_in);
        size_t tb = 0;
        BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, dkeys_in, dkeys_out, ncand, 0, 20, st));
        BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
        BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(ctx->cub_tmp.p, tb, dkeys_in, dkeys_out, ncand, 0, 20, st));
        d_order = dkeys_out;
      }
      (void)hipEventRecord(ctx->ev[6], st);
      // In-launch re-growth of a plane that lost a point: always for short lists.  For long lists only in rounds
      // with many attempts: such a round lasts as long as its longest independent plane, so a long plane that
      // is grown again has the time to finish inside the same launch (urban 10 M: +20 %); in a round of a few
      // chained planes (the facade) the re-growth only repeats work the next round does anyway (-23 %).
      const int rml = (ncand >= retry_big_round && !retry_env) ? 0x7fffffff : retry_max_list;
      const bool grow_v1 = grow_force >= 0 ? grow_force == 0 : !(KC == 16 && ncand < 65536);
      if (grow_v1) {
        if (KC == 16)
          grow_spec_kernel<16><<<ncand, 64, lds_pad, st>>>(a, d_cand, ncand, rec, dead, pool, d_out, 512 * n + 4096, rml, d_order);
        else
          grow_spec_kernel<32><<<ncand, 64, lds_pad, st>>>(a, d_cand, ncand, rec, dead, pool, d_out, 512 * n + 4096, rml, d_order);
      } else {
        if (KC == 16)
          grow_spec2_kernel<16><<<ncand, 64, lds_pad, st>>>(a, d_cand, ncand, rec, dead, pool, d_out, 512 * n + 4096, rml, d_order);
        else
          grow_spec2_kernel<32><<<ncand, 64, lds_pad, st>>>(a, d_cand, ncand, rec, dead, pool, d_out, 512 * n + 4096, rml, d_order);
      }
      (void)hipEventRecord(ctx->ev[7], st);
      grow_launches++;
      timed_round = true;
      if (forge_mode) {  // self-test: corrupt one finished plane (see forge_kernel); once per call
        forge_kernel<<<1, 64, 0, st>>>(d_out, ncand, pool.base, forge_mode, d_misc + 5);
        forge_mode = 0;
      }
      validate1_kernel<<<ncand, VT, 0, st>>>(d_out, ncand, pool.base, rec, quads, dead, n, vmark,
                                             (int32_t)((rounds & 0x3ff) << 20), d_misc + 4);
      if (do_validate3) {
        // beside the owner passes: it only reads the finished lists and the records' geometry, and is one
        // wave of sequential f64 adds per plane -- the main stream's kernels are memory bound
        BS_HIP(ctx, hipEventRecord(ctx->sev[0], st));
        BS_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->sev[0], 0));
        validate3_kernel<<<ncand, V3T, 0, ctx->side>>>(d_out, ncand, pool.base, rec, quads, d_misc + 4);
        BS_HIP(ctx, hipEventRecord(ctx->sev[1], ctx->side));
        v3_pending = true;
      }
      reset_tags_kernel<<<ncand, VT, 0, st>>>(d_out, ncand, pool.base, rec, quads, K);
      // dead[] is only ever written for seeds of this round's attempts: clear exactly those (not 4 n bytes per round)
      reset_dead_kernel<<<nblk(ncand, 256), 256, 0, st>>>(d_cand, ncand, dead);
      // insert the finished planes and let the owners settle
      plane_apply_kernel<<<ncand, VT, 0, st>>>(d_out, ncand, pool.base, base, ps, dcur, bcur, omega, occ, rec, quads);
      rc = propagate();
      if (rc != BS_OK)
        return rc;
      validate2_kernel<<<ncand, VT, 0, st>>>(d_out, ncand, pool.base, omega, rec, quads, K);
    }
    if (npend)
      validate2_kernel<<<npend, VT, 0, st>>>(d_pend, npend, pstore, omega, rec, quads, K);
    // lowest candidate under the new owners (= first attempt that is not established yet) and, in the
    // same pass, the candidate list of the next round
    const int32_t init2[2] = {0, INF};
    BS_HIP(ctx, hipMemcpyAsync(d_misc + 1, init2, sizeof init2, hipMemcpyHostToDevice, st));
    cand_flag_kernel<<<(int)((nb256 + cand_sub - 1) / cand_sub), 256, 0, st>>>(rec, quads, K, n, F, prio, ps, 0, omega, d_misc + 2, cand_raw, d_misc + 1, bcand, cand_sub);
    // (d_misc[1] = number of candidates listed, d_misc[2] = the lowest one: one copy into the page-locked flags)
    BS_HIP(ctx, hipMemcpyAsync(h_flags + 8, d_misc + 1, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (v3_pending) {  // validate3's verdicts (PlaneOut.v3ok, the refusal counter) are read below
      BS_HIP(ctx, hipStreamWaitEvent(st, ctx->sev[1], 0));
      v3_pending = false;
    }
    BS_HIP(ctx, hipMemcpyAsync(h_flags + 10, d_misc + 4, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // A big round's attempts mostly failed at depth 0 (152 k of 158 k in the first round of the 50 M cloud) and the
    // host never looks at those: only the others travel, in seed order (20 MB each way and a 2 ms walk over
    // page-locked memory otherwise).  nh = entries on the host, d_outh = the device array they mirror.
    const bool compact = ncand >= 4096 && !dbg;
    int nh = ncand;
    PlaneOut* d_outh = d_out;
    if (compact) {
      size_t tb = 0;
      BS_HIP(ctx, hipcub::DeviceSelect::If(nullptr, tb, d_out, d_outc, d_misc + 20, ncand, KeepForHost(), st));
      BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
      BS_HIP(ctx, hipcub::DeviceSelect::If(ctx->cub_tmp.p, tb, d_out, d_outc, d_misc + 20, ncand, KeepForHost(), st));
      BS_HIP(ctx, hipMemcpyAsync(h_flags + 14, d_misc + 20, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      nh = h_flags[14];
      d_outh = d_outc;
    }
    if (nh)
      BS_HIP(ctx, hipMemcpyAsync(h_out, d_outh, sizeof(PlaneOut) * nh, hipMemcpyDeviceToHost, st));
    if (npend)
      BS_HIP(ctx, hipMemcpyAsync(h_pend, d_pend, sizeof(PlaneOut) * npend, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    const int32_t next_ncand_all = h_flags[8], new_min = h_flags[9], rejects_now = h_flags[10];
    if (timed_round) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess)
        grow_ms += ms;
      timed_round = false;
    }
    // first attempt (by seed index) whose result is not established
    int32_t first_bad = new_min;
    bool nomem_lowest = false;
    for (int w = 0; w < nh; w++) {
      const PlaneOut& o = h_out[w];
      if (dbg && o.list_n > n + 1)
        fprintf(stderr, "[bs] IMPOSSIBLE list: round %ld w=%d seed=%d status=%d consistent=%d list_n=%ld steps=%ld log=%ld thief=%d\n",
                (long)rounds, w, o.seed, o.status, o.consistent, (long)o.list_n, (long)o.steps, (long)o.log_n, o.thief);
      if (o.status == ST_WATCHDOG)
        return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): watchdog");
      if (o.status == ST_DONE && (!o.consistent || !o.v3ok))
        first_bad = std::min(first_bad, o.seed);
      if (o.status == ST_DONE && !o.consistent) {  // which test of validate2 failed (PlaneOut.pad4: 1 seed row, 2 list, 4 log)
        incons[0] += (o.pad4 & 1) ? 1 : 0;
        incons[1] += (o.pad4 & 2) ? 1 : 0;
        incons[2] += (o.pad4 & 4) ? 1 : 0;
      }
      if (o.w == 0 && o.status == ST_NOMEM)
        nomem_lowest = true;
      if (o.status == ST_NOMEM) {
        full_refresh = true;  // its last claims may be neither listed nor reset
        dead_dirty = true;
      }
    }
    for (int w = 0; w < npend; w++)
      if (!h_pend[w].consistent)
        first_bad = std::min(first_bad, h_pend[w].seed);
    if (nomem_lowest) {
      if (max_waves == 1)
        return fail(ctx, BS_ERR_NOMEM, "region grow (speculative): round pool exhausted");
      max_waves = std::max(1, max_waves / 8);
    }
#ifdef BS_PROBE
    if (dbg) {
      unsigned long long hp[32];
      (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_prof2), sizeof(hp));
      const double nstep = (double)hp[10] + 1.0;
      fprintf(stderr, "[prof2] steps=%llu expansions=%llu | pop+issue=%.0f state=%.0f wait=%.0f test+classify=%.0f claims+walk=%.0f log+counters=%.0f "
                      "flushchk=%.0f list+defer=%.0f push=%.0f top=%.0f (cycles per step, cumulative over rounds)\n",
              hp[10], hp[11], hp[0] / nstep, hp[1] / nstep, hp[2] / nstep, hp[3] / nstep, hp[4] / nstep, hp[5] / nstep, hp[6] / nstep,
              hp[7] / nstep, hp[8] / nstep, hp[9] / nstep);
    }
#endif
    if (dbg) {
      int cnt[6] = {0, 0, 0, 0, 0, 0}, cons = 0, pcons = 0;
      int64_t maxsteps = 0, sumsteps = 0, maxlist = 0;
      for (int w = 0; w < ncand; w++) {
        cnt[h_out[w].status]++;
        cons += h_out[w].status == ST_DONE && h_out[w].consistent;
        maxsteps = std::max(maxsteps, h_out[w].steps);
        sumsteps += h_out[w].steps;
        maxlist = std::max(maxlist, h_out[w].list_n);
      }
      for (int w = 0; w < npend; w++)
        pcons += h_pend[w].consistent;
      {  // when did the longest attempts start and end, relative to the first wave of the launch?
        int64_t t0 = INT64_MAX, t1 = 0;
        std::vector<int> idx;
        for (int w = 0; w < ncand; w++) {
          t0 = std::min(t0, h_out[w].t_start);
          t1 = std::max(t1, h_out[w].t_end);
          if (h_out[w].steps > 2000)
            idx.push_back(w);
        }
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return h_out[x].steps > h_out[y].steps; });
        fprintf(stderr, "[bs]   launch span %.2f ms; longest attempts (w, steps, status, start ms, end ms, us/step):", (t1 - t0) / 1e5);
        for (size_t q = 0; q < std::min<size_t>(idx.size(), 12); q++) {
          const PlaneOut& o = h_out[idx[q]];
          fprintf(stderr, " [%d %ld %d %.2f %.2f %.3f xcc%d se%d cu%d simd%d]", idx[q], (long)o.steps, o.status, (o.t_start - t0) / 1e5,
                  (o.t_end - t0) / 1e5, (o.t_end - o.t_start) / 100.0 / (double)o.steps, (o.pad5 >> 16) & 0xf, (o.pad5 >> 13) & 7,
                  (o.pad5 >> 8) & 0xf, (o.pad5 >> 4) & 3);
        }
        fprintf(stderr, "\n");
        if (ncand > 20000) {
          fprintf(stderr, "[bs]   start ms by w:");
          for (int w = 0; w < ncand; w += 8192)
            fprintf(stderr, " %d:%.2f", w, (h_out[w].t_start - t0) / 1e5);
          double life0 = 0, lifep = 0;
          int64_t n0 = 0, npl = 0;
          for (int w = 0; w < ncand; w++) {
            if (h_out[w].status == ST_FAILED0) {
              life0 += (h_out[w].t_end - h_out[w].t_start) / 100.0;
              n0++;
            } else {
              lifep += (h_out[w].t_end - h_out[w].t_start) / 100.0;
              npl++;
            }
          }
          fprintf(stderr, "\n[bs]   mean lifetime: failed0 %.1f us (%ld), others %.1f us (%ld); wave-time total %.1f ms\n", life0 / std::max<int64_t>(n0, 1),
                  (long)n0, lifep / std::max<int64_t>(npl, 1), (long)npl, (life0 + lifep) / 1e3);
        }
      }
      int why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int w = 0; w < ncand; w++)
        if (h_out[w].status == ST_DONE && !h_out[w].consistent)
          why[h_out[w].pad4 & 7]++;
      fprintf(stderr, "[bs]   inconsistent new planes by failed test (1 seed row, 2 list, 4 log; sums): 1:%d 2:%d 3:%d 4:%d 5:%d 6:%d 7:%d\n",
              why[1], why[2], why[3], why[4], why[5], why[6], why[7]);
      fprintf(stderr,
              "[bs] round %ld F=%d ncand_all=%d grown=%d done=%d (consistent %d) failed0=%d nomem=%d stolen=%d "
              "pending=%d (consistent %d) first_bad=%d new_min=%d maxsteps=%ld sumsteps=%ld maxlist=%ld passes=%ld\n",
              (long)rounds, F, ncand_all, ncand, cnt[ST_DONE], cons, cnt[ST_FAILED0], cnt[ST_NOMEM], cnt[ST_STOLEN],
              npend, pcons, first_bad, new_min, (long)maxsteps, (long)sumsteps, (long)maxlist, (long)passes);
      if (getenv("BS_DEBUG_PLANES"))
        for (int w = 0; w < ncand; w++)
          fprintf(stderr, "[bs]   plane seed=%d status=%d consistent=%d steps=%ld list=%ld log=%ld thief=%d\n",
                  h_out[w].seed, h_out[w].status, h_out[w].consistent, (long)h_out[w].steps, (long)h_out[w].list_n,
                  (long)h_out[w].log_n, h_out[w].thief);
    }
    // merge pending and new planes in seed order: below first_bad -> final,
    // consistent ones above it stay pending, the rest is dropped from the structure
    std::vector<PlaneOut> next_pending;
    int finals = 0, dropped = 0;
    int pend_room = MAX_PENDING;  // old pending planes that stay have priority over new ones
    for (int w = 0; w < npend; w++)
      pend_room -= (h_pend[w].consistent && h_pend[w].seed >= first_bad) ? 1 : 0;
    {
      int ip = 0, iw = 0;
      while (ip < npend || iw < nh) {
        const bool take_p = (iw >= nh) || (ip < npend && h_pend[ip].seed < h_out[iw].seed);
        PlaneOut& o = take_p ? h_pend[ip] : h_out[iw];
        const int32_t* src = take_p ? pstore : pool.base;
        if (take_p)
          ip++;
        else
          iw++;
        o.pad = 0;  // command for plane_apply_kernel: 2 = drop
        if (o.status != ST_DONE)
          continue;  // never entered the structure
        if (!o.consistent || !o.v3ok) {
          o.pad = 2;
          dropped++;
          continue;  // re-enters as a candidate (or orphan maker) next round
        }
        if (o.seed < first_bad) {
          finals++;
          if (dbg && o.keep)
            fprintf(stderr, "[bs]   commit seed=%d n=%ld log=%ld steps=%ld from=%s round=%ld first_bad=%d\n", o.seed,
                    (long)o.list_n, (long)o.log_n, (long)o.steps, take_p ? "pending" : "new", (long)rounds, first_bad);
          rc = commit_plane(o, src);
          if (rc != BS_OK)
            return rc;
        } else if (take_p) {
          next_pending.push_back(o);
        } else if (pend_room > 0 && pstore_top + o.list_n + o.log_n + 8 <= pstore_cap) {
          pend_room--;
          PlaneOut q = o;
          q.list_off = pstore_top;
          copies.push_back({pool.base + o.list_off, pstore + q.list_off, o.list_n, nullptr});
          pstore_top += (o.list_n + 3) & ~(int64_t)3;
          q.log_off = pstore_top;
          if (o.log_n)
            copies.push_back({pool.base + o.log_off, pstore + q.log_off, o.log_n, nullptr});
          pstore_top += (o.log_n + 3) & ~(int64_t)3;
          next_pending.push_back(q);
        } else {
          o.pad = 2;  // no room to keep it: drop, it will be grown again
          dropped++;
        }
      }
    }
    rc = flush_copies();
    if (rc != BS_OK)
      return rc;
    if (dropped) {
      if (nh) {
        BS_HIP(ctx, hipMemcpyAsync(d_outh, h_out, sizeof(PlaneOut) * nh, hipMemcpyHostToDevice, st));
        plane_apply_kernel<<<nh, VT, 0, st>>>(d_outh, nh, pool.base, base, ps, dcur, bcur, omega, occ, rec, quads);
      }
      if (npend) {
        BS_HIP(ctx, hipMemcpyAsync(d_pend, h_pend, sizeof(PlaneOut) * npend, hipMemcpyHostToDevice, st));
        plane_apply_kernel<<<npend, VT, 0, st>>>(d_pend, npend, pstore, base, ps, dcur, bcur, omega, occ, rec, quads);
      }
      rc = propagate();
      if (rc != BS_OK)
        return rc;
    } else {
      cand_listed = true;  // owners unchanged since the candidate pass above: its list is the next round's
      ncand_all = next_ncand_all;
    }
    pending.swap(next_pending);
    if (pending.empty())
      pstore_top = 0;
    if (first_bad == INF)
      break;  // every plane attempt is final; omega holds the remaining orphan makers' fixed point
    // a plane the validation refused is simply grown again; only a refusal that repeats is fatal
    const bool refused = rejects_now > rejects_seen;
    rejects_seen = rejects_now;
    refused_rounds = refused ? refused_rounds + 1 : 0;
    if (finals == 0 && dropped == 0 && !nomem_lowest && first_bad <= F && (!refused || refused_rounds > 8))
      return fail(ctx, BS_ERR_INTERNAL, refused ? "region grow (speculative): the validation keeps refusing a plane"
                                                : "region grow (speculative): no progress");
    F = first_bad;
    if (rounds > 4 * n + 16)
      return fail(ctx, BS_ERR_INTERNAL, "region grow (speculative): round limit");
  }
  int32_t* owner_final = omega;
  // labels and plane records
  const int np = (int)recs.size();
  if (np > 0) {
    BS_HIP(ctx, hipMemcpyAsync(d_seeds, seeds.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, st));
    BS_HIP(ctx, hipMemcpyAsync(ctx->rg_planes.p, recs.data(), sizeof(PlaneRec) * np, hipMemcpyHostToDevice, st));
  }
  label_kernel<<<nblk(n, 256), 256, 0, st>>>(owner_final, n, d_seeds, np, prio, d_plane_idx);
  int32_t vstat[12] = {0};
  BS_HIP(ctx, hipMemcpyAsync(vstat, d_misc + 4, sizeof vstat, hipMemcpyDeviceToHost, st));
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
  ctx->rg_omega = owner_final;
  ctx->rg_prio = prio;
  ctx->rg_seeds = d_seeds;
  ctx->rg_nplanes = np;
  ctx->tm.largest_plane = largest;
  ctx->tm.n_seed_attempts = attempts;
  ctx->tm.rg_rounds = rounds;
  ctx->tm.grow_kernel_ms = grow_ms;
  ctx->tm.grow_kernel_launches = grow_launches;
  ctx->tm.validation_rejects = vstat[0];
  ctx->tm.forged_seed = vstat[1] - 1;
  ctx->tm.forged_refused = vstat[2];
  ctx->tm.rej_v1_robbed = vstat[8];
  ctx->tm.rej_v1_tag = vstat[9];
  ctx->tm.rej_v1_dup = vstat[10];
  ctx->tm.rej_v3_state = vstat[11];
  ctx->tm.incons_seed = incons[0];
  ctx->tm.incons_list = incons[1];
  ctx->tm.incons_log = incons[2];
  {
    float ms = 0.f;
    ctx->tm.grow_setup_ms = hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[9]) == hipSuccess ? (double)ms : 0.0;
  }
  ctx->tm.audit_attempts = -1;
  ctx->tm.audit_mismatches = 0;
  ctx->tm.audit_ms = 0.0;
  if (ctx->audit || getenv("BS_AUDIT")) {
    // Every plane attempt that exists under the final owners, grown again without speculation (a.F = INF:
    // an owner below the seed is final and taken, everything else is free at the seed's time) and compared
    // with what was committed -- see bs_set_audit in include/bs_api.h.
    const auto t0 = std::chrono::steady_clock::now();
    BS_HIP(ctx, hipMemsetAsync(bcand, 1, nb256, st));
    BS_HIP(ctx, hipMemsetAsync(d_misc + 1, 0, sizeof(int32_t), st));
    BS_HIP(ctx, hipMemsetAsync(d_misc + 8, 0, 2 * sizeof(int32_t), st));
    cand_flag_kernel<<<(int)((nb256 + cand_sub - 1) / cand_sub), 256, 0, st>>>(rec, quads, K, n, 0, prio, ps, 1, omega, nullptr,
                                                                              cand_raw, d_misc + 1, bcand, cand_sub);
    BS_HIP(ctx, hipMemcpyAsync(h_flags + 11, d_misc + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    const int na = h_flags[11];
    // Committed planes only ever claim their own points, so all of them replay concurrently.  An attempt the
    // reference rolled back gave its points free again: its replay may claim points that belong to later
    // planes (or to another rolled-back attempt), so these are replayed one at a time on clean tags.
    std::vector<unsigned long long> hc((size_t)na), hc_sorted;
    if (na > 0)
      BS_HIP(ctx, hipMemcpy(hc.data(), cand_raw, sizeof(unsigned long long) * na, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    int n_committed = 0;
    {
      std::vector<unsigned long long> rest;
      for (const unsigned long long c : hc) {
        const int32_t sd = (int32_t)(c >> 32);
        if (std::binary_search(seeds.begin(), seeds.end(), sd))
          hc_sorted.push_back(c);
        else
          rest.push_back(c);
      }
      n_committed = (int)hc_sorted.size();
      hc_sorted.insert(hc_sorted.end(), rest.begin(), rest.end());
    }
    if (na > 0)
      BS_HIP(ctx, hipMemcpyAsync(d_cand, hc_sorted.data(), sizeof(unsigned long long) * na, hipMemcpyHostToDevice, st));
    a.F = INF;
    auto grow_n = [&](int off, int cnt, PlaneOut* o) {
      const bool grow_v1 = grow_force >= 0 ? grow_force == 0 : !(KC == 16 && cnt < 65536);
      if (grow_v1) {
        if (KC == 16)
          grow_spec_kernel<16><<<cnt, 64, 0, st>>>(a, d_cand + off, cnt, rec, dead, pool, o, 512 * n + 4096, 0, nullptr);
        else
          grow_spec_kernel<32><<<cnt, 64, 0, st>>>(a, d_cand + off, cnt, rec, dead, pool, o, 512 * n + 4096, 0, nullptr);
      } else {
        if (KC == 16)
          grow_spec2_kernel<16><<<cnt, 64, 0, st>>>(a, d_cand + off, cnt, rec, dead, pool, o, 512 * n + 4096, 0, nullptr);
        else
          grow_spec2_kernel<32><<<cnt, 64, 0, st>>>(a, d_cand + off, cnt, rec, dead, pool, o, 512 * n + 4096, 0, nullptr);
      }
    };
    auto compare_n = [&](int cnt, int32_t* retry) {
      audit_compare_kernel<<<cnt, VT, 0, st>>>(d_out, cnt, pool.base, prio, d_seeds, ctx->rg_planes.as<PlaneRec>(), np,
                                               ctx->rg_list.as<int32_t>(), a.th_count, d_misc + 8, dead, retry);
    };
    for (int off = 0; off < n_committed; off += wave_cap) {
      const int cnt = std::min(n_committed - off, wave_cap);
      BS_HIP(ctx, hipMemsetAsync(d_pool_top, 0, sizeof(unsigned long long), st));
      grow_n(off, cnt, d_out);
      compare_n(cnt, nullptr);
      reset_tags_kernel<<<cnt, VT, 0, st>>>(d_out, cnt, pool.base, rec, quads, K);
    }
    // Attempts the reference rolled back (:203-208): a few hundred points each.  They gave their points free again, so
    // two of them (or one and a later plane) may want the same point: all of them are replayed CONCURRENTLY, and those
    // that met another attempt of the batch -- robbed, or one logged "held by an earlier in-flight attempt" -- are
    // replayed again in the next batch, on clean tags.  The lowest seed of a batch always comes through, so the
    // batches shrink; in practice one or two suffice (one launch per attempt before: 0.6 s at 50 M, 3 s at 200 M).
    {
      std::vector<unsigned long long> pend(hc_sorted.begin() + n_committed, hc_sorted.end()), next;
      int32_t* d_retry = reinterpret_cast<int32_t*>(dkeys_in);  // (free outside the rounds: wave_cap words)
      std::vector<int32_t> h_retry;
      int batches = 0;
      while (!pend.empty()) {
        next.clear();
        for (size_t off = 0; off < pend.size(); off += (size_t)wave_cap) {
          const int cnt = (int)std::min<size_t>(pend.size() - off, (size_t)wave_cap);
          BS_HIP(ctx, hipMemcpyAsync(d_cand, pend.data() + off, sizeof(unsigned long long) * cnt, hipMemcpyHostToDevice, st));
          BS_HIP(ctx, hipMemsetAsync(d_pool_top, 0, sizeof(unsigned long long), st));
          reset_dead_kernel<<<nblk(cnt, 256), 256, 0, st>>>(d_cand, cnt, dead);
          grow_n(0, cnt, d_out);
          compare_n(cnt, d_retry);
          reset_tags_kernel<<<cnt, VT, 0, st>>>(d_out, cnt, pool.base, rec, quads, K);
          reset_dead_kernel<<<nblk(cnt, 256), 256, 0, st>>>(d_cand, cnt, dead);
          h_retry.resize((size_t)cnt);
          BS_HIP(ctx, hipMemcpyAsync(h_retry.data(), d_retry, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
          BS_HIP(ctx, hipStreamSynchronize(st));
          for (int i = 0; i < cnt; i++)
            if (h_retry[i])
              next.push_back(pend[off + i]);
        }
        if (next.size() == pend.size())
          return fail(ctx, BS_ERR_INTERNAL, "audit: a batch of rolled-back attempts made no progress");
        pend.swap(next);
        batches++;
      }
      if (dbg)
        fprintf(stderr, "[bs] audit: %d rolled-back attempts replayed in %d batches\n", na - n_committed, batches);
    }
    BS_HIP(ctx, hipMemcpyAsync(h_flags + 12, d_misc + 8, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    BS_HIP(ctx, hipGetLastError());
    ctx->tm.audit_attempts = na;
    ctx->tm.audit_mismatches = (int64_t)h_flags[12] + (h_flags[13] != np ? 1 : 0);
    ctx->tm.audit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (dbg || ctx->tm.audit_mismatches)
      fprintf(stderr, "[bs] audit: %d attempts replayed, %d of %d committed planes met, %d mismatches, %.1f ms\n", na, h_flags[13], np,
              h_flags[12], ctx->tm.audit_ms);
  }
  return BS_OK;
}

}  // namespace bs
