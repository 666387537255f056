// bs_grow.hip -- region-growing plane labelling (gfx950).  Product code.
//
// Replaces seg_plane::get_planes / seg_plane::Broad
// (/root/reference/tmc3/my_function.cpp:180-258).  The reference is a strictly
// sequential, order-dependent greedy DFS with a running plane estimate; its
// labels are reproduced bit for bit:
//   * one 64-lane wavefront executes one Broad() call per step: lanes 0..K-2
//     each gather one neighbour (label, xyz, normal) and evaluate the
//     thickness / normal tests against the wave-uniform plane state;
//   * __ballot + prefix popcount give the ordered compaction of the accepted
//     neighbours (selectedId order == lane order == neighbour-list order);
//   * the plane state uses running sums extended in append order (f64 normal
//     sum, wrapping uint32 centre sum) -- bit-identical to the reference's
//     O(|plane|) re-summation at every call (Appendix B.4);
//   * recursion is an explicit LIFO in HBM; the first accepted child is kept
//     in registers, the others are pushed in reverse.
// rg_mode 1 (this file's grow_seq_kernel) runs the whole seed scan on one
// wavefront and is trivially exact; it is also the validation baseline of the
// speculative multi-plane scheduler.
#include "bs_common.h"

namespace bs {

namespace {

struct GrowArgs {
  const int32_t* xyz;
  const double* normals;
  const int32_t* neigh;
  int64_t n;
  int K;
  double th;        // (double)th_thickness
  double cos_th;
  int64_t th_count;
  int32_t* plane_idx;
  int32_t* list_pool;
  int64_t list_cap;
  int32_t* stack;
  int64_t stack_cap;
  PlaneRec* planes;
  int32_t planes_cap;
  GrowStats* stats;
  int64_t step_cap;
};

// loads that must observe this wave's own earlier stores: served from L2
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

__global__ __launch_bounds__(64) void grow_seq_kernel(GrowArgs a)
{
  const int lane = threadIdx.x;
  const int K = a.K;
  const int nc = K - 1;
  int64_t list_off = 0;
  int np = 0;
  int cur_id = 1;  // my_function.h:119
  int64_t attempts = 0, largest = 0, steps = 0;
  int err = 0;
  int64_t base = 0;
  while (base < a.n && !err) {
    const int64_t i = base + lane;
    const int lab = (i < a.n) ? ld_i32(a.plane_idx + i) : 0;
    const unsigned long long fm = __ballot(lab == -1);  // my_function.cpp:185
    if (fm == 0) {
      base += 64;
      continue;
    }
    const int64_t seed = base + (__ffsll(fm) - 1);
    base = seed + 1;
    attempts++;
    // :187-191 -- the seed enters pointIdx but is NOT labelled (Q1)
    double cnx = a.normals[3 * seed], cny = a.normals[3 * seed + 1], cnz = a.normals[3 * seed + 2];
    int ccx = a.xyz[3 * seed], ccy = a.xyz[3 * seed + 1], ccz = a.xyz[3 * seed + 2];
    double Sx = 0.0 + cnx, Sy = 0.0 + cny, Sz = 0.0 + cnz;  // re-summed from {0,0,0} in :241-248
    uint32_t Cx = (uint32_t)ccx, Cy = (uint32_t)ccy, Cz = (uint32_t)ccz;
    int64_t ln = 1;
    if (list_off + 1 > a.list_cap) {
      err = 1;
      break;
    }
    if (lane == 0)
      a.list_pool[list_off] = (int32_t)seed;
    int64_t sp = 0;
    int64_t cur = seed;
    bool depth0 = true, failed = false;
    for (;;) {
      if (++steps > a.step_cap) {
        err = 3;
        break;
      }
      // ---- one Broad(cur, depth) call, :220-258 ----
      const bool act = lane < nc;
      int cand = 0, clab = 1, px = 0, py = 0, pz = 0;
      double mx = 0, my = 0, mz = 0;
      if (act) {
        cand = a.neigh[cur * K + lane + 1];  // slot 0 skipped (Q5)
        clab = ld_i32(a.plane_idx + cand);
        px = a.xyz[3 * (int64_t)cand];
        py = a.xyz[3 * (int64_t)cand + 1];
        pz = a.xyz[3 * (int64_t)cand + 2];
        mx = a.normals[3 * (int64_t)cand];
        my = a.normals[3 * (int64_t)cand + 1];
        mz = a.normals[3 * (int64_t)cand + 2];
      }
      bool ok = false;
      if (act && clab <= 0) {  // Q4
        const int dx = (int)((uint32_t)px - (uint32_t)ccx);
        const int dy = (int)((uint32_t)py - (uint32_t)ccy);
        const int dz = (int)((uint32_t)pz - (uint32_t)ccz);
        const double dist = __builtin_fabs((double)dx * cnx + (double)dy * cny + (double)dz * cnz);
        const double dt = cnx * mx + cny * my + cnz * mz;
        ok = dist <= a.th && dt >= a.cos_th;  // :230
      }
      unsigned long long am = __ballot(ok);
      {
        // a caller-supplied row may repeat an index: the reference labels the point on first sight and finds it labelled
        // on the second (:226-233), so only the LOWEST accepting lane of an index keeps it
        bool dup = false;
        unsigned long long mm = am;
        while (mm) {
          const int l = __ffsll(mm) - 1;
          mm &= mm - 1;
          const int cl = readlane_i32(cand, l);
          dup = dup || (l < lane && cl == cand);
        }
        ok = ok && !dup;
        am = __ballot(ok);
      }
      const int cnt = __popcll(am);
      if (ok)
        a.plane_idx[cand] = cur_id;  // :233
      if (depth0 && cnt < nc) {  // :238-239 (Q2: labels stay)
        failed = true;
        break;
      }
      depth0 = false;
      if (cnt) {
        if (list_off + ln + cnt > a.list_cap || sp + cnt > a.stack_cap) {
          err = 2;
          break;
        }
        const int rank = __popcll(am & ((1ull << lane) - 1ull));
        if (ok)
          a.list_pool[list_off + ln + rank] = cand;  // :232
        // running sums in append order (== selectedId order)
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
        // :249-250
        const double nrm = __builtin_sqrt((Sx * Sx) + (Sy * Sy) + (Sz * Sz));
        cnx = Sx / nrm;
        cny = Sy / nrm;
        cnz = Sz / nrm;
        const uint64_t dn = (uint64_t)ln;
        ccx = (int32_t)((uint64_t)(int64_t)(int32_t)Cx / dn);  // int /= size_t (Q3)
        ccy = (int32_t)((uint64_t)(int64_t)(int32_t)Cy / dn);
        ccz = (int32_t)((uint64_t)(int64_t)(int32_t)Cz / dn);
        // :252-255 -- first child continues in registers, the rest go on the
        // stack in reverse so that they pop in selection order
        if (ok && rank > 0)
          a.stack[sp + (cnt - 1 - rank)] = cand;
        sp += cnt - 1;
        cur = readlane_i32(cand, __ffsll(am) - 1);
      } else {
        if (sp == 0)
          break;
        sp--;
        cur = ld_i32(a.stack + sp);
      }
    }
    if (err)
      break;
    if (failed)
      continue;  // :193-194
    if (ln > a.th_count) {  // :199
      if (np >= a.planes_cap) {
        err = 4;
        break;
      }
      if (lane == 0) {
        PlaneRec r;
        r.normal[0] = cnx;
        r.normal[1] = cny;
        r.normal[2] = cnz;
        r.center[0] = ccx;
        r.center[1] = ccy;
        r.center[2] = ccz;
        r.list_off = list_off;
        r.list_n = ln;
        r.id = cur_id;
        r.seed = (int32_t)seed;
        r.pad = 0;
        a.planes[np] = r;
      }
      np++;
      cur_id++;
      list_off += ln;
      largest = ln > largest ? ln : largest;
    } else {
      for (int64_t t = lane; t < ln; t += 64)  // :203-208
        a.plane_idx[ld_i32(a.list_pool + list_off + t)] = -1;
    }
  }
  if (lane == 0) {
    GrowStats s;
    s.n_planes = np;
    s.error = err;
    s.list_used = list_off;
    s.seed_attempts = attempts;
    s.largest = largest;
    s.steps = steps;
    *a.stats = s;
  }
}

}  // namespace

int launch_region_grow_seq(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals, const int32_t* d_neigh,
                       int64_t n, const bs_params& p, int32_t* d_plane_idx)
{
  hipStream_t st = ctx->stream;
  ctx->rg_valid = false;
  ctx->rg_omega = nullptr;  // (the single-wave grower keeps no owner array: bs_owner_fetch_dev refuses)
  ctx->rg_prio = nullptr;
  ctx->rg_seeds = nullptr;
  ctx->rg_nplanes = 0;
  const int64_t list_cap = 2 * n + 64;
  const int64_t stack_cap = n + 64;
  const int64_t planes_cap = n / std::max(1, p.th_point_count) + 64;
  BS_HIP(ctx, ctx->rg_list.reserve(sizeof(int32_t) * list_cap));
  BS_HIP(ctx, ctx->rg_stack.reserve(sizeof(int32_t) * stack_cap));
  BS_HIP(ctx, ctx->rg_planes.reserve(sizeof(PlaneRec) * planes_cap));
  BS_HIP(ctx, ctx->rg_stats.reserve(sizeof(GrowStats)));
  BS_HIP(ctx, hipMemsetAsync(d_plane_idx, 0xFF, sizeof(int32_t) * n, st));  // my_function.h:103
  BS_HIP(ctx, hipMemsetAsync(ctx->rg_stats.p, 0, sizeof(GrowStats), st));
  GrowArgs a;
  a.xyz = d_xyz;
  a.normals = d_normals;
  a.neigh = d_neigh;
  a.n = n;
  a.K = p.k;
  a.th = (double)p.th_thickness;
  a.cos_th = p.cos_th;
  a.th_count = p.th_point_count;
  a.plane_idx = d_plane_idx;
  a.list_pool = ctx->rg_list.as<int32_t>();
  a.list_cap = list_cap;
  a.stack = ctx->rg_stack.as<int32_t>();
  a.stack_cap = stack_cap;
  a.planes = ctx->rg_planes.as<PlaneRec>();
  a.planes_cap = (int32_t)std::min<int64_t>(planes_cap, INT32_MAX);
  a.stats = ctx->rg_stats.as<GrowStats>();
  a.step_cap = 512 * n + 4096;
  (void)hipEventRecord(ctx->ev[6], st);
  grow_seq_kernel<<<1, 64, 0, st>>>(a);
  (void)hipEventRecord(ctx->ev[7], st);
  BS_HIP(ctx, hipGetLastError());
  GrowStats hs;
  BS_HIP(ctx, hipMemcpyAsync(&hs, ctx->rg_stats.p, sizeof hs, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  if (hs.error)
    return fail(ctx, BS_ERR_INTERNAL, "region grow: pool overflow or watchdog");
  ctx->rg_n = n;
  ctx->rg_valid = true;
  ctx->tm.largest_plane = hs.largest;
  ctx->tm.n_seed_attempts = hs.seed_attempts;
  ctx->tm.rg_rounds = 1;
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess)
      ctx->tm.grow_kernel_ms = ms;
    ctx->tm.grow_kernel_launches = 1;
    ctx->tm.grow_setup_ms = 0.0;
    ctx->tm.validation_rejects = 0;
    ctx->tm.forged_seed = -1;
    ctx->tm.forged_refused = 0;
    ctx->tm.audit_attempts = -1;  // the sequential baseline needs no audit
    ctx->tm.audit_mismatches = 0;
    ctx->tm.audit_ms = 0.0;
  }
  return BS_OK;
}

}  // namespace bs
