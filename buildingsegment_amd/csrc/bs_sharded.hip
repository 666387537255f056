// bs_sharded.hip -- bs_segment_sharded: ONE cloud segmented by all ranks of a communicator (gfx950 + RCCL).
// Product code; the C++-host twin of buildingsegment_amd/dist.py (same steps, same device kernels).
//
// Reference loops this replaces for a host that runs one process (or thread) per GPU: the normal / kNN loops of
// /root/reference/tmc3/my_function.h:63,71-78 and the ordered seed scan of /root/reference/tmc3/my_function.cpp:184-217.
//
//   1. Morton partition      63-bit keys of the own points, sorted (rocPRIM radix sort as a primitive); splitters
//                            from all-gathered samples; ONE split-size all-to-all of 16-byte rows (x, y, z, gidx)
//   2. halo                  every rank publishes the voxels (edge >= h) its slab occupies (sorted, all-gathered);
//                            a point goes to peer r iff one of the 27 voxels around its own is occupied by r
//   3. slab kNN + normals    bs_knn_normals_dev with global tie-breaking; certified iff k-th distance < h, else the
//                            ranks agree (all-reduce MAX) on a doubled halo
//   4. components            bs_cc_hook_dev + all-reduce(MIN) of the parent array until no rank hooks
//   5. deal + redistribute   (root, count) lists all-gathered, dealt on the host (same answer on every rank),
//                            ONE all-to-all per array (rows, k-lists, normals)
//   6. localize + grow       sort by global index, k-lists renumbered (bs_remap_rows_dev), bs_region_grow_dev
//   7. global labels         committed seeds all-gathered + sorted, labels from owners, all-reduce(MAX)
// A failure on one rank is agreed on before the next collective (the flag rides on the phase's own all-reduce):
// every rank returns an error, none is left waiting.
#include <hipcub/hipcub.hpp>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "bs_common.h"

namespace bs {

namespace {

inline int nblk(int64_t n, int b) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + b - 1) / b, 0x7fffffff)); }

__host__ __device__ inline uint64_t spread21s(uint64_t v)
{
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

struct I3 {
  int64_t x, y, z;
};

__global__ void pack_rows_kernel(const int32_t* __restrict__ xyz, const int32_t* __restrict__ gidx, int64_t m, int4* __restrict__ rows)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < m)
    rows[i] = make_int4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], gidx ? gidx[i] : (int32_t)i);
}

__global__ void unpack_rows_kernel(const int4* __restrict__ rows, int64_t m, int32_t* __restrict__ xyz, int32_t* __restrict__ gidx)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= m)
    return;
  const int4 r = rows[i];
  xyz[3 * i] = r.x;
  xyz[3 * i + 1] = r.y;
  xyz[3 * i + 2] = r.z;
  if (gidx)
    gidx[i] = r.w;
}

__global__ void morton_key_kernel(const int4* __restrict__ rows, int64_t m, I3 mn, int shift, uint64_t* __restrict__ keys,
                                  int32_t* __restrict__ vals)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= m)
    return;
  const int4 r = rows[i];
  const uint64_t x = (uint64_t)((int64_t)r.x - mn.x) >> shift, y = (uint64_t)((int64_t)r.y - mn.y) >> shift,
                 z = (uint64_t)((int64_t)r.z - mn.z) >> shift;
  keys[i] = spread21s(x) | (spread21s(y) << 1) | (spread21s(z) << 2);
  vals[i] = (int32_t)i;
}

__global__ void gather_int4_kernel(const int4* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int4* __restrict__ dst)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[idx[i]];
}

// rows of `words` 32-bit words gathered by index (k-lists: k words, normals: 6 words)
__global__ void gather_rows_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int words,
                                   int32_t* __restrict__ dst)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n * words)
    return;
  const int64_t i = t / words;
  const int w = (int)(t - i * words);
  dst[t] = src[(int64_t)idx[i] * words + w];
}

__global__ void sample_kernel(const uint64_t* __restrict__ skeys, int64_t m, int S, uint64_t* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S)
    return;
  out[i] = m > 0 ? skeys[((int64_t)i * (m - 1)) / (S > 1 ? S - 1 : 1)] : ~0ull;  // (i < 2^10, m < 2^31)
}

// pos[j] = number of sorted keys <= splitters[j]  (destination boundaries of the sorted order)
__global__ void upper_bound_kernel(const uint64_t* __restrict__ skeys, int64_t m, const uint64_t* __restrict__ spl, int ns,
                                   int64_t* __restrict__ pos)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns)
    return;
  const uint64_t s = spl[j];
  int64_t lo = 0, hi = m;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skeys[mid] <= s)
      lo = mid + 1;
    else
      hi = mid;
  }
  pos[j] = lo;
}

__device__ inline uint64_t voxel_key(int4 r, I3 org, int v)
{
  // floor division (coordinates are >= origin: the global minimum) + 1: room for the -1 offsets
  const uint64_t vx = (uint64_t)(((int64_t)r.x - org.x) / v) + 1, vy = (uint64_t)(((int64_t)r.y - org.y) / v) + 1,
                 vz = (uint64_t)(((int64_t)r.z - org.z) / v) + 1;
  return (vx << 42) | (vy << 21) | vz;
}

__global__ void voxel_key_kernel(const int4* __restrict__ rows, int64_t n, I3 org, int v, uint64_t* __restrict__ keys)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    keys[i] = voxel_key(rows[i], org, v);
}

__global__ void halo_mark_kernel(const uint64_t* __restrict__ vkeys, int64_t n, const uint64_t* __restrict__ occ, int64_t cnt,
                                 uint8_t* __restrict__ flags)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint64_t k0 = vkeys[i];
  bool hit = false;
  for (int dx = -1; dx <= 1 && !hit; dx++)
    for (int dy = -1; dy <= 1 && !hit; dy++)
      for (int dz = -1; dz <= 1 && !hit; dz++) {
        const uint64_t q = k0 + ((uint64_t)(int64_t)dx << 42) + ((uint64_t)(int64_t)dy << 21) + (uint64_t)(int64_t)dz;
        int64_t lo = 0, hi = cnt;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (occ[mid] < q)
            lo = mid + 1;
          else
            hi = mid;
        }
        hit = lo < cnt && occ[lo] == q;
      }
  flags[i] = hit ? 1 : 0;
}

__global__ void iota_kernel(int32_t* p, int64_t n)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = (int32_t)i;
}

__global__ void fill_kernel(int32_t* p, int64_t n, int32_t v)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = v;
}

// out[i] = idx[i] >= 0 ? table[idx[i]] : -1
__global__ void lookup_kernel(const int32_t* __restrict__ table, const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ out)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = idx[i] >= 0 ? table[idx[i]] : -1;
}

__global__ void row_gidx_kernel(const int4* __restrict__ rows, int64_t n, int32_t* __restrict__ g, int32_t* __restrict__ iota)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  g[i] = rows[i].w;
  if (iota)
    iota[i] = (int32_t)i;
}

__global__ void dest_key_kernel(const int32_t* __restrict__ root, int64_t n, const int32_t* __restrict__ uniq,
                                const int32_t* __restrict__ dest, int nu, uint32_t* __restrict__ key, int32_t* __restrict__ val)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t r = root[i];
  int lo = 0, hi = nu;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (uniq[mid] < r)
      lo = mid + 1;
    else
      hi = mid;
  }
  key[i] = (uint32_t)dest[lo];
  val[i] = (int32_t)i;
}

__global__ void count_dest_kernel(const uint32_t* __restrict__ skey, int64_t n, int world, int64_t* __restrict__ pos)
{
  // pos[r] = number of sorted destination keys < r + 1
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= world)
    return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skey[mid] <= (uint32_t)r)
      lo = mid + 1;
    else
      hi = mid;
  }
  pos[r] = lo;
}

__global__ void dup_check_kernel(const int32_t* __restrict__ sg, int64_t n, int* bad)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i + 1 < n && sg[i] == sg[i + 1])
    *bad = 1;
}

__global__ void scatter_labels_kernel(const int32_t* __restrict__ sg, const int32_t* __restrict__ lab, int64_t n, int32_t* __restrict__ full)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n)
    full[sg[i]] = lab[i];
}

__global__ void reduce_i32_kernel(int32_t* dst, const int32_t* src, int64_t n, int op)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t a = dst[i], b = src[i];
  dst[i] = op == BS_MIN ? (a < b ? a : b) : (op == BS_MAX ? (a > b ? a : b) : a + b);
}

__global__ void reduce_i64_kernel(int64_t* dst, const int64_t* src, int64_t n, int op)
{
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int64_t a = dst[i], b = src[i];
  dst[i] = op == BS_MIN ? (a < b ? a : b) : (op == BS_MAX ? (a > b ? a : b) : a + b);
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
};

Rccl& rccl()
{
  static Rccl r = [] {
    Rccl q;
    // the copy already in the process (torch's, under bench.py) wins; otherwise ROCm's
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      q.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (q.lib)
        break;
    }
    if (!q.lib)
      return q;
    auto sym = [&](const char* s) { return dlsym(q.lib, s); };
    q.GetUniqueId = (decltype(q.GetUniqueId))sym("ncclGetUniqueId");
    q.CommInitRank = (decltype(q.CommInitRank))sym("ncclCommInitRank");
    q.CommDestroy = (decltype(q.CommDestroy))sym("ncclCommDestroy");
    q.AllReduce = (decltype(q.AllReduce))sym("ncclAllReduce");
    q.AllGather = (decltype(q.AllGather))sym("ncclAllGather");
    q.Send = (decltype(q.Send))sym("ncclSend");
    q.Recv = (decltype(q.Recv))sym("ncclRecv");
    q.GroupStart = (decltype(q.GroupStart))sym("ncclGroupStart");
    q.GroupEnd = (decltype(q.GroupEnd))sym("ncclGroupEnd");
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.AllGather && q.Send && q.Recv && q.GroupStart &&
           q.GroupEnd;
    return q;
  }();
  return r;
}

struct RcclHandle {
  ncclComm_t comm;
  int rank, world;
};

int rccl_all_reduce(void* h, void* d_buf, int64_t count, int dtype, int op, void* stream)
{
  auto* c = (RcclHandle*)h;
  const ncclRedOp_t ro = op == BS_MIN ? ncclMin : (op == BS_MAX ? ncclMax : ncclSum);
  return rccl().AllReduce(d_buf, d_buf, (size_t)count, dtype == BS_I64 ? ncclInt64 : ncclInt32, ro, c->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

int rccl_all_gather(void* h, const void* d_send, void* d_recv, int64_t bytes, void* stream)
{
  auto* c = (RcclHandle*)h;
  return rccl().AllGather(d_send, d_recv, (size_t)bytes, ncclChar, c->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

int rccl_all_to_all_v(void* h, const void* d_send, const int64_t* sb, void* d_recv, const int64_t* rb, void* stream)
{
  // split-size all-to-all as ONE group of point-to-point transfers: xGMI is point to point, every pair of GPUs has
  // its own link, and only what a peer needs crosses it
  auto* c = (RcclHandle*)h;
  Rccl& R = rccl();
  if (R.GroupStart() != ncclSuccess)
    return 1;
  int64_t so = 0, ro = 0;
  bool bad = false;
  // (a block travels in pieces of at most 1 GiB: a 3.2 GB payload came back incomplete from ONE all-to-all over RCCL
  // through torch.distributed -- byte counts beyond 2^31; sender and receiver cut a block at the same places)
  const int64_t piece = (int64_t)1 << 30;
  for (int r = 0; r < c->world; r++) {
    for (int64_t o = 0; o < sb[r]; o += piece)
      bad = bad || R.Send((const char*)d_send + so + o, (size_t)std::min(piece, sb[r] - o), ncclChar, r, c->comm, (hipStream_t)stream) != ncclSuccess;
    for (int64_t o = 0; o < rb[r]; o += piece)
      bad = bad || R.Recv((char*)d_recv + ro + o, (size_t)std::min(piece, rb[r] - o), ncclChar, r, c->comm, (hipStream_t)stream) != ncclSuccess;
    so += sb[r];
    ro += rb[r];
  }
  if (R.GroupEnd() != ncclSuccess)
    return 1;
  return bad ? 1 : 0;
}

// ---- in-process communicator: the ranks are threads of one process ---------------------------------------------
// (one host thread per GPU without a launcher, or several ranks sharing one GPU in the tests).  Buffers are
// exchanged by device-to-device copies between the ranks' allocations, synchronised by a host barrier.
struct LocalShared {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t epoch = 0;
  std::vector<const void*> ptr;
  std::vector<const int64_t*> cnt;
  std::vector<int> dev;
  std::atomic<int> refs;
  explicit LocalShared(int w) : world(w), ptr(w), cnt(w), dev(w), refs(w) {}
  void barrier()
  {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t e = epoch;
    if (++arrived == world) {
      arrived = 0;
      epoch++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return epoch != e; });
    }
  }
};

struct LocalHandle {
  LocalShared* sh;
  int rank;
};

int local_all_reduce(void* h, void* d_buf, int64_t count, int dtype, int op, void* stream)
{
  auto* c = (LocalHandle*)h;
  LocalShared* S = c->sh;
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamSynchronize(st) != hipSuccess)
    return 1;
  S->ptr[c->rank] = d_buf;
  S->barrier();
  int rc = 0;
  if (c->rank == 0) {  // rank 0 folds every peer's buffer into its own ...
    for (int r = 1; r < S->world && !rc; r++) {
      if (dtype == BS_I64)
        reduce_i64_kernel<<<nblk(count, 256), 256, 0, st>>>((int64_t*)d_buf, (const int64_t*)S->ptr[r], count, op);
      else
        reduce_i32_kernel<<<nblk(count, 256), 256, 0, st>>>((int32_t*)d_buf, (const int32_t*)S->ptr[r], count, op);
    }
    rc = hipStreamSynchronize(st) != hipSuccess;
  }
  S->barrier();
  if (c->rank != 0) {  // ... and the peers copy the result
    const size_t bytes = (size_t)count * (dtype == BS_I64 ? 8 : 4);
    rc = hipMemcpyAsync(d_buf, S->ptr[0], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess;
  }
  S->barrier();
  return rc;
}

int local_all_gather(void* h, const void* d_send, void* d_recv, int64_t bytes, void* stream)
{
  auto* c = (LocalHandle*)h;
  LocalShared* S = c->sh;
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamSynchronize(st) != hipSuccess)
    return 1;
  S->ptr[c->rank] = d_send;
  S->barrier();
  int rc = 0;
  for (int r = 0; r < S->world && !rc; r++)
    if (bytes > 0)
      rc = hipMemcpyAsync((char*)d_recv + (size_t)r * bytes, S->ptr[r], (size_t)bytes, hipMemcpyDeviceToDevice, st) != hipSuccess;
  rc = rc || hipStreamSynchronize(st) != hipSuccess;
  S->barrier();
  return rc;
}

int local_all_to_all_v(void* h, const void* d_send, const int64_t* sb, void* d_recv, const int64_t* rb, void* stream)
{
  auto* c = (LocalHandle*)h;
  LocalShared* S = c->sh;
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamSynchronize(st) != hipSuccess)
    return 1;
  S->ptr[c->rank] = d_send;
  S->cnt[c->rank] = sb;
  S->barrier();
  int rc = 0;
  int64_t ro = 0;
  for (int r = 0; r < S->world && !rc; r++) {
    int64_t so = 0;  // offset of MY block in rank r's send buffer
    for (int q = 0; q < c->rank; q++)
      so += S->cnt[r][q];
    if (S->cnt[r][c->rank] != rb[r])
      rc = 1;
    else if (rb[r] > 0)
      rc = hipMemcpyAsync((char*)d_recv + ro, (const char*)S->ptr[r] + so, (size_t)rb[r], hipMemcpyDeviceToDevice, st) != hipSuccess;
    ro += rb[r];
  }
  rc = rc || hipStreamSynchronize(st) != hipSuccess;
  S->barrier();
  return rc;
}

// ---- the sharded pass ------------------------------------------------------------------------------------------
enum Sh { ROWS, KEYS_A, KEYS_B, VALS_A, VALS_B, OWN, SMALL, SEND, HALO, LOC_XYZ, LOC_GIDX, NEIGH, NORMALS, PARENT, ROOT, FLAGS,
          G_OWN, G_NG, G_NR, SG, L_XYZ, L_NG, L_NR, L_MISC };

struct Pass {
  bs_ctx* ctx;
  const bs_comm_ops* cm;
  hipStream_t st;
  int rank, world;
  template <class T>
  T* buf(int which, size_t count)
  {
    if (ctx->sh[which].reserve(sizeof(T) * std::max<size_t>(count, 1)) != hipSuccess)
      return nullptr;
    return ctx->sh[which].as<T>();
  }
  bool live() const { return cm != nullptr; }
  int all_reduce(void* d, int64_t count, int dtype, int op)
  {
    if (!cm)
      return BS_OK;
    return cm->all_reduce(cm->handle, d, count, dtype, op, st) ? fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: all-reduce failed") : BS_OK;
  }
  int all_gather(const void* s, void* r, int64_t bytes)
  {
    if (!cm) {
      BS_HIP(ctx, hipMemcpyAsync(r, s, (size_t)bytes, hipMemcpyDeviceToDevice, st));
      return BS_OK;
    }
    return cm->all_gather(cm->handle, s, r, bytes, st) ? fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: all-gather failed") : BS_OK;
  }
  // host vector all-reduce through a small device buffer
  int all_reduce_host(int64_t* v, int n, int op)
  {
    int64_t* d = buf<int64_t>(SMALL, 64) + 32;
    BS_HIP(ctx, hipMemcpyAsync(d, v, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
    int rc = all_reduce(d, n, BS_I64, op);
    if (rc != BS_OK)
      return rc;
    BS_HIP(ctx, hipMemcpyAsync(v, d, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    return BS_OK;
  }
  // every rank's vector of n int64 (host) -> matrix [world][n] on the host
  int all_gather_host(const int64_t* v, int n, std::vector<int64_t>& out)
  {
    out.assign((size_t)world * n, 0);
    int64_t* d = buf<int64_t>(SMALL, 64 + (size_t)(world + 1) * n + 64) + 64;
    BS_HIP(ctx, hipMemcpyAsync(d, v, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
    int rc = all_gather(d, d + n, (int64_t)sizeof(int64_t) * n);
    if (rc != BS_OK)
      return rc;
    BS_HIP(ctx, hipMemcpyAsync(out.data(), d + n, sizeof(int64_t) * n * world, hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    return BS_OK;
  }
  // rows of row_bytes sorted by destination; send_rows[world] (host).  Received rows land in ctx->sh[dst_buf];
  // *n_recv = their number.
  int all_to_all_rows(const void* d_send, const std::vector<int64_t>& send_rows, int64_t row_bytes, int dst_buf, void** d_recv,
                      int64_t* n_recv)
  {
    std::vector<int64_t> mat;
    int rc = all_gather_host(send_rows.data(), world, mat);
    if (rc != BS_OK)
      return rc;
    std::vector<int64_t> sb(world), rb(world);
    int64_t tot = 0;
    for (int r = 0; r < world; r++) {
      sb[r] = send_rows[r] * row_bytes;
      rb[r] = mat[(size_t)r * world + rank] * row_bytes;
      tot += mat[(size_t)r * world + rank];
    }
    char* recv = buf<char>(dst_buf, (size_t)tot * row_bytes + 16);
    if (!recv)
      return fail(ctx, BS_ERR_NOMEM, "bs_segment_sharded: device allocation failed");
    if (cm) {
      if (cm->all_to_all_v(cm->handle, d_send, sb.data(), recv, rb.data(), st))
        return fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: all-to-all failed");
    } else if (tot > 0) {
      BS_HIP(ctx, hipMemcpyAsync(recv, d_send, (size_t)tot * row_bytes, hipMemcpyDeviceToDevice, st));
    }
    *d_recv = recv;
    *n_recv = tot;
    return BS_OK;
  }
  // agreement on failures: the local status rides on an all-reduce(MAX); every rank leaves with an error if any failed
  int agree(int local_rc, const char* what)
  {
    int64_t f = local_rc != BS_OK ? 1 : 0;
    std::string keep = ctx->err;
    int rc = all_reduce_host(&f, 1, BS_MAX);
    if (rc != BS_OK)
      return rc;
    if (local_rc != BS_OK) {
      ctx->err = keep;
      return local_rc;
    }
    if (f)
      return fail(ctx, BS_ERR_INTERNAL, what);
    return BS_OK;
  }
};

double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// deal of the components (same algorithm as buildingsegment_amd/dist.py::assign_components)
void assign_components(const std::vector<int64_t>& roots, const std::vector<int64_t>& counts, const std::vector<int>& ranks, int world,
                       std::vector<int32_t>& uniq, std::vector<int32_t>& dest)
{
  std::vector<int64_t> u(roots);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  const size_t nu = u.size();
  std::vector<int64_t> per(nu * world, 0), size(nu, 0);
  for (size_t i = 0; i < roots.size(); i++) {
    const size_t c = std::lower_bound(u.begin(), u.end(), roots[i]) - u.begin();
    per[c * world + ranks[i]] += counts[i];
    size[c] += counts[i];
  }
  std::vector<int> home(nu);
  for (size_t c = 0; c < nu; c++)
    home[c] = (int)(std::max_element(per.begin() + c * world, per.begin() + (c + 1) * world) - (per.begin() + c * world));
  std::vector<size_t> order(nu);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return size[a] != size[b] ? size[a] > size[b] : u[a] < u[b]; });
  const size_t exact_top = 8192;
  std::vector<int64_t> load(world, 0);
  dest.assign(nu, 0);
  int64_t total = 0;
  for (size_t c = 0; c < nu; c++) {
    dest[c] = home[c];
    total += size[c];
  }
  for (size_t t = exact_top; t < nu; t++)
    load[home[order[t]]] += size[order[t]];
  const double cap = 1.10 * (double)total / world;
  for (size_t t = 0; t < std::min(exact_top, nu); t++) {
    const size_t c = order[t];
    const int h = home[c];
    const int least = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    const double lim = std::max(cap, (double)(load[least] + size[c]));
    const int d = (double)(load[h] + size[c]) <= lim ? h : least;
    dest[c] = d;
    load[d] += size[c];
  }
  uniq.resize(nu);
  for (size_t c = 0; c < nu; c++)
    uniq[c] = (int32_t)u[c];
}

#define SH_CHECK(p)                                                                  \
  do {                                                                               \
    if (!(p))                                                                        \
      return fail(ctx, BS_ERR_NOMEM, "bs_segment_sharded: device allocation failed"); \
  } while (0)

int sort_pairs_u64(bs_ctx* ctx, const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, int64_t n, int end_bit)
{
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, end_bit, ctx->stream));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tb, kin, kout, vin, vout, (int)n, 0, end_bit, ctx->stream));
  return BS_OK;
}

int sort_pairs_u32(bs_ctx* ctx, const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout, int64_t n, int end_bit)
{
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, end_bit, ctx->stream));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tb, kin, kout, vin, vout, (int)n, 0, end_bit, ctx->stream));
  return BS_OK;
}

int sort_pairs_i32(bs_ctx* ctx, const int32_t* kin, int32_t* kout, const int32_t* vin, int32_t* vout, int64_t n)
{
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, 32, ctx->stream));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tb, kin, kout, vin, vout, (int)n, 0, 32, ctx->stream));
  return BS_OK;
}

template <class K>
int sort_keys(bs_ctx* ctx, const K* kin, K* kout, int64_t n)
{
  size_t tb = 0;
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, kin, kout, (int)n, 0, (int)sizeof(K) * 8, ctx->stream));
  BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
  BS_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(ctx->cub_tmp.p, tb, kin, kout, (int)n, 0, (int)sizeof(K) * 8, ctx->stream));
  return BS_OK;
}

int sharded_impl(bs_ctx* ctx, const bs_comm_ops* comm, const int32_t* d_xyz, const int32_t* d_gidx, int64_t m, int64_t n_total,
                 const bs_params& p, double halo, int32_t* d_plane_idx, bs_shard_info* info)
{
  // BS_SHARD_FORCE_COMM=1: issue every collective even at world size 1 (the RCCL calls on a one-GPU box)
  const bool use_comm = comm && (comm->world > 1 || getenv("BS_SHARD_FORCE_COMM") != nullptr);
  Pass P{ctx, use_comm ? comm : nullptr, ctx->stream, comm ? comm->rank : 0, comm ? comm->world : 1};
  hipStream_t st = ctx->stream;
  const int world = P.world, rank = P.rank, k = p.k;
  bs_shard_info inf;
  memset(&inf, 0, sizeof inf);
  ctx->sh_valid = false;
  // small scalars and staging areas live in ONE buffer that is sized once (a later, larger request would move it
  // under the pointers handed out before)
  SH_CHECK(P.buf<int64_t>(SMALL, 8192 + (size_t)(world + 4) * 1100));
  double t0 = now_ms();
  auto lap = [&](double& slot) {
    (void)hipStreamSynchronize(st);
    const double t = now_ms();
    slot += t - t0;
    t0 = t;
  };

  // ---- 1. Morton partition ----
  int4* rows = P.buf<int4>(ROWS, (size_t)m);
  SH_CHECK(rows);
  if (m > 0)
    pack_rows_kernel<<<nblk(m, 256), 256, 0, st>>>(d_xyz, d_gidx, m, rows);
  int32_t bb[6];
  int rc = bbox_dev(ctx, d_xyz, m, bb);
  if (rc != BS_OK)
    return rc;
  int64_t mnmx[6];
  for (int a = 0; a < 3; a++) {
    mnmx[a] = m > 0 ? bb[a] : ((int64_t)1 << 40);
    mnmx[3 + a] = m > 0 ? -(int64_t)bb[3 + a] : ((int64_t)1 << 40);  // max as min of the negation: ONE all-reduce
  }
  rc = P.all_reduce_host(mnmx, 6, BS_MIN);
  if (rc != BS_OK)
    return rc;
  const I3 origin = {mnmx[0], mnmx[1], mnmx[2]};
  int64_t ext = 0;
  for (int a = 0; a < 3; a++)
    ext = std::max(ext, -mnmx[3 + a] - mnmx[a]);
  int shift = 0;
  while ((ext >> shift) >= (1 << 21))
    shift++;
  int4* own = rows;
  int64_t n_own = m;
  if (P.live()) {
    uint64_t* ka = P.buf<uint64_t>(KEYS_A, (size_t)m);
    uint64_t* kb = P.buf<uint64_t>(KEYS_B, (size_t)m);
    int32_t* va = P.buf<int32_t>(VALS_A, (size_t)m);
    int32_t* vb = P.buf<int32_t>(VALS_B, (size_t)m);
    int4* srt = P.buf<int4>(SEND, (size_t)m);
    SH_CHECK(ka && kb && va && vb && srt);
    if (m > 0) {
      morton_key_kernel<<<nblk(m, 256), 256, 0, st>>>(rows, m, origin, shift, ka, va);
      rc = sort_pairs_u64(ctx, ka, kb, va, vb, m, 63);
      if (rc != BS_OK)
        return rc;
      gather_int4_kernel<<<nblk(m, 256), 256, 0, st>>>(rows, vb, m, srt);
    }
    const int S = 1024;
    uint64_t* d_s = (uint64_t*)P.buf<int64_t>(SMALL, 64 + (size_t)(world + 2) * S + 64) + 64;
    SH_CHECK(d_s);
    sample_kernel<<<nblk(S, 256), 256, 0, st>>>(kb, m, S, d_s);
    rc = P.all_gather(d_s, d_s + S, (int64_t)sizeof(uint64_t) * S);
    if (rc != BS_OK)
      return rc;
    std::vector<uint64_t> samp((size_t)world * S);
    BS_HIP(ctx, hipMemcpyAsync(samp.data(), d_s + S, sizeof(uint64_t) * samp.size(), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    std::sort(samp.begin(), samp.end());
    const int64_t valid = std::lower_bound(samp.begin(), samp.end(), ~0ull) - samp.begin();
    std::vector<uint64_t> spl(world - 1, 0);
    for (int r = 1; r < world; r++)
      spl[r - 1] = valid ? samp[(size_t)std::min<int64_t>(valid - 1, std::max<int64_t>(0, valid * r / world))] : 0;
    BS_HIP(ctx, hipMemcpyAsync(d_s, spl.data(), sizeof(uint64_t) * (world - 1), hipMemcpyHostToDevice, st));
    int64_t* d_pos = (int64_t*)(d_s + world);
    upper_bound_kernel<<<1, 256, 0, st>>>(kb, m, d_s, world - 1, d_pos);
    std::vector<int64_t> pos(world, m);
    BS_HIP(ctx, hipMemcpyAsync(pos.data(), d_pos, sizeof(int64_t) * (world - 1), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    pos[world - 1] = m;
    std::vector<int64_t> send(world);
    for (int r = 0; r < world; r++)
      send[r] = pos[r] - (r ? pos[r - 1] : 0);
    void* got = nullptr;
    rc = P.all_to_all_rows(srt, send, sizeof(int4), OWN, &got, &n_own);
    if (rc != BS_OK)
      return rc;
    own = (int4*)got;
  }
  inf.n_own = n_own;
  lap(inf.ms_partition);

  // ---- 2 + 3. halo exchange, slab kNN + normals, certification ----
  double h = halo > 0 ? halo : 2.0 * p.radius;
  h = std::max(h, p.radius);
  int32_t* neigh = P.buf<int32_t>(NEIGH, (size_t)n_own * k);
  double* normals = P.buf<double>(NORMALS, (size_t)n_own * 3);
  int32_t* gidx_own = nullptr;
  SH_CHECK(neigh && normals);
  int retries = 0;
  for (;;) {
    int64_t n_halo = 0;
    int4* halo_rows = nullptr;
    if (P.live()) {
      const int v = (int)std::max(std::ceil(h), 500.0);
      uint64_t* vk = P.buf<uint64_t>(KEYS_A, (size_t)n_own);
      uint64_t* vs = P.buf<uint64_t>(KEYS_B, (size_t)n_own + 64);
      SH_CHECK(vk && vs);
      int64_t nocc = 0;
      int32_t* d_cnt = (int32_t*)P.buf<int64_t>(SMALL, 64);
      if (n_own > 0) {
        voxel_key_kernel<<<nblk(n_own, 256), 256, 0, st>>>(own, n_own, origin, v, vk);
        rc = sort_keys<uint64_t>(ctx, vk, vs, n_own);
        if (rc != BS_OK)
          return rc;
        uint64_t* occ_tmp = (uint64_t*)P.buf<int32_t>(VALS_A, 2 * (size_t)n_own + 16);
        SH_CHECK(occ_tmp);
        size_t tb = 0;
        BS_HIP(ctx, hipcub::DeviceSelect::Unique(nullptr, tb, vs, occ_tmp, d_cnt, (int)n_own, st));
        BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
        BS_HIP(ctx, hipcub::DeviceSelect::Unique(ctx->cub_tmp.p, tb, vs, occ_tmp, d_cnt, (int)n_own, st));
        int32_t c32 = 0;
        BS_HIP(ctx, hipMemcpyAsync(&c32, d_cnt, sizeof c32, hipMemcpyDeviceToHost, st));
        BS_HIP(ctx, hipStreamSynchronize(st));
        nocc = c32;
      }
      std::vector<int64_t> cnts;
      rc = P.all_gather_host(&nocc, 1, cnts);
      if (rc != BS_OK)
        return rc;
      const int64_t cap = std::max<int64_t>(1, *std::max_element(cnts.begin(), cnts.end()));
      uint64_t* occ_pad = (uint64_t*)P.buf<int32_t>(VALS_B, 2 * (size_t)cap * (world + 1) + 16);
      SH_CHECK(occ_pad);
      BS_HIP(ctx, hipMemsetAsync(occ_pad, 0xff, sizeof(uint64_t) * cap, st));
      if (nocc > 0)
        BS_HIP(ctx, hipMemcpyAsync(occ_pad, ctx->sh[VALS_A].p, sizeof(uint64_t) * nocc, hipMemcpyDeviceToDevice, st));
      uint64_t* occ_all = occ_pad + cap;
      rc = P.all_gather(occ_pad, occ_all, (int64_t)sizeof(uint64_t) * cap);
      if (rc != BS_OK)
        return rc;
      // per peer: mark, compact into the send buffer
      uint8_t* flags = P.buf<uint8_t>(FLAGS, (size_t)n_own + 16);
      int4* send = P.buf<int4>(SEND, (size_t)n_own * (size_t)std::max(1, world - 1) + 16);
      SH_CHECK(flags && send);
      std::vector<int64_t> scount(world, 0);
      int64_t soff = 0;
      for (int r = 0; r < world; r++) {
        if (r == rank || n_own == 0 || cnts[r] == 0)
          continue;
        halo_mark_kernel<<<nblk(n_own, 256), 256, 0, st>>>(vk, n_own, occ_all + (size_t)r * cap, cnts[r], flags);
        size_t tb = 0;
        BS_HIP(ctx, hipcub::DeviceSelect::Flagged(nullptr, tb, own, flags, send + soff, d_cnt, (int)n_own, st));
        BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
        BS_HIP(ctx, hipcub::DeviceSelect::Flagged(ctx->cub_tmp.p, tb, own, flags, send + soff, d_cnt, (int)n_own, st));
        int32_t c32 = 0;
        BS_HIP(ctx, hipMemcpyAsync(&c32, d_cnt, sizeof c32, hipMemcpyDeviceToHost, st));
        BS_HIP(ctx, hipStreamSynchronize(st));
        scount[r] = c32;
        soff += c32;
      }
      void* got = nullptr;
      rc = P.all_to_all_rows(send, scount, sizeof(int4), HALO, &got, &n_halo);
      if (rc != BS_OK)
        return rc;
      halo_rows = (int4*)got;
    }
    const int64_t nl = n_own + n_halo;
    inf.n_local = nl;
    int32_t* lxyz = P.buf<int32_t>(LOC_XYZ, (size_t)nl * 3);
    int32_t* lg = P.buf<int32_t>(LOC_GIDX, (size_t)nl);
    SH_CHECK(lxyz && lg);
    if (n_own > 0)
      unpack_rows_kernel<<<nblk(n_own, 256), 256, 0, st>>>(own, n_own, lxyz, lg);
    if (n_halo > 0)
      unpack_rows_kernel<<<nblk(n_halo, 256), 256, 0, st>>>(halo_rows, n_halo, lxyz + 3 * n_own, lg + n_own);
    gidx_own = lg;  // the first n_own entries
    lap(inf.ms_halo);
    int64_t unc = 0;
    int lrc = BS_OK;
    if (n_own > 0 && nl >= k)
      lrc = bs_knn_normals_dev(ctx, lxyz, lg, nl, 0, n_own, &p, neigh, normals, P.live() ? h : 0.0, &unc);
    else
      unc = n_own;  // a slab that cannot fill one k-list: force a wider halo
    int64_t flag[2] = {unc, lrc != BS_OK ? 1 : 0};
    const std::string keep = ctx->err;
    rc = P.all_reduce_host(flag, 2, BS_MAX);
    if (rc != BS_OK)
      return rc;
    lap(inf.ms_knn);
    if (lrc != BS_OK) {
      ctx->err = keep;
      return lrc;
    }
    if (flag[1])
      return fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: another rank failed in kNN + normals");
    if (flag[0] == 0)
      break;
    if (++retries > 6 || !P.live())
      return fail(ctx, BS_ERR_UNCERTIFIED, "bs_segment_sharded: the halo exchange could not certify every k-list");
    h *= 2.0;
  }
  inf.halo_mm = h;
  inf.halo_retries = retries;

  // ---- 4. connected components ----
  int4* g_own = own;
  int32_t* g_ng = neigh;
  double* g_nr = normals;
  int64_t n_loc = n_own;
  if (P.live()) {
    int32_t* parent = P.buf<int32_t>(PARENT, (size_t)n_total);
    SH_CHECK(parent);
    iota_kernel<<<nblk(n_total, 256), 256, 0, st>>>(parent, n_total);
    for (int it = 1;; it++) {
      int64_t hooks = 0;
      const int lrc = bs_cc_hook_dev(ctx, neigh, gidx_own, n_own, k, parent, n_total, &hooks);
      rc = P.all_reduce(parent, n_total, BS_I32, BS_MIN);  // the union-find all-reduce over xGMI
      if (rc != BS_OK)
        return rc;
      int64_t flag[2] = {hooks, lrc != BS_OK ? 1 : 0};
      const std::string keep = ctx->err;
      rc = P.all_reduce_host(flag, 2, BS_SUM);
      if (rc != BS_OK)
        return rc;
      if (lrc != BS_OK) {
        ctx->err = keep;
        return lrc;
      }
      if (flag[1])
        return fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: another rank failed in the connected components");
      inf.cc_iterations = it;
      if (flag[0] == 0)
        break;
      if (it >= 64)
        return fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: the union-find did not settle");
    }
    int32_t* root = P.buf<int32_t>(ROOT, (size_t)n_own + 16);
    SH_CHECK(root);
    if (n_own > 0)
      lookup_kernel<<<nblk(n_own, 256), 256, 0, st>>>(parent, gidx_own, n_own, root);
    lap(inf.ms_components);

    // ---- 5. deal + redistribution ----
    int32_t* rs = P.buf<int32_t>(VALS_A, (size_t)n_own + 16);
    int32_t* ur = P.buf<int32_t>(VALS_B, 2 * (size_t)n_own + 16);
    SH_CHECK(rs && ur);
    int32_t* uc = ur + n_own + 8;
    int64_t nruns = 0;
    int32_t* d_cnt = (int32_t*)P.buf<int64_t>(SMALL, 64);
    if (n_own > 0) {
      rc = sort_keys<int32_t>(ctx, root, rs, n_own);
      if (rc != BS_OK)
        return rc;
      size_t tb = 0;
      BS_HIP(ctx, hipcub::DeviceRunLengthEncode::Encode(nullptr, tb, rs, ur, uc, d_cnt, (int)n_own, st));
      BS_HIP(ctx, ctx->cub_tmp.reserve(tb));
      BS_HIP(ctx, hipcub::DeviceRunLengthEncode::Encode(ctx->cub_tmp.p, tb, rs, ur, uc, d_cnt, (int)n_own, st));
      int32_t c32 = 0;
      BS_HIP(ctx, hipMemcpyAsync(&c32, d_cnt, sizeof c32, hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      nruns = c32;
    }
    std::vector<int64_t> cnts;
    rc = P.all_gather_host(&nruns, 1, cnts);
    if (rc != BS_OK)
      return rc;
    const int64_t cap = std::max<int64_t>(1, *std::max_element(cnts.begin(), cnts.end()));
    // (root, count) pairs as int32 x 2, padded to cap per rank
    int32_t* pad = P.buf<int32_t>(KEYS_A, 2 * (size_t)cap * (world + 1) + 16);
    SH_CHECK(pad);
    BS_HIP(ctx, hipMemsetAsync(pad, 0, sizeof(int32_t) * 2 * cap, st));
    if (nruns > 0) {
      BS_HIP(ctx, hipMemcpyAsync(pad, ur, sizeof(int32_t) * nruns, hipMemcpyDeviceToDevice, st));
      BS_HIP(ctx, hipMemcpyAsync(pad + cap, uc, sizeof(int32_t) * nruns, hipMemcpyDeviceToDevice, st));
    }
    rc = P.all_gather(pad, pad + 2 * cap, (int64_t)sizeof(int32_t) * 2 * cap);
    if (rc != BS_OK)
      return rc;
    std::vector<int32_t> hall((size_t)2 * cap * world);
    BS_HIP(ctx, hipMemcpyAsync(hall.data(), pad + 2 * cap, sizeof(int32_t) * hall.size(), hipMemcpyDeviceToHost, st));
    BS_HIP(ctx, hipStreamSynchronize(st));
    std::vector<int64_t> troot, tcnt;
    std::vector<int> trank;
    for (int r = 0; r < world; r++)
      for (int64_t j = 0; j < cnts[r]; j++) {
        troot.push_back(hall[(size_t)r * 2 * cap + j]);
        tcnt.push_back(hall[(size_t)r * 2 * cap + cap + j]);
        trank.push_back(r);
      }
    std::vector<int32_t> uniq, dest;
    assign_components(troot, tcnt, trank, world, uniq, dest);
    inf.components = (int64_t)uniq.size();
    const int nu = (int)uniq.size();
    int32_t* d_uniq = P.buf<int32_t>(KEYS_B, 2 * (size_t)nu + 16);
    SH_CHECK(d_uniq);
    int32_t* d_dest = d_uniq + nu + 4;
    if (nu) {
      BS_HIP(ctx, hipMemcpyAsync(d_uniq, uniq.data(), sizeof(int32_t) * nu, hipMemcpyHostToDevice, st));
      BS_HIP(ctx, hipMemcpyAsync(d_dest, dest.data(), sizeof(int32_t) * nu, hipMemcpyHostToDevice, st));
    }
    uint32_t* dk = (uint32_t*)P.buf<int32_t>(VALS_A, 2 * (size_t)n_own + 32);
    int32_t* dv = P.buf<int32_t>(VALS_B, 2 * (size_t)n_own + 32);
    SH_CHECK(dk && dv);
    uint32_t* dks = dk + n_own + 8;
    int32_t* dvs = dv + n_own + 8;
    std::vector<int64_t> send(world, 0);
    if (n_own > 0) {
      dest_key_kernel<<<nblk(n_own, 256), 256, 0, st>>>(root, n_own, d_uniq, d_dest, nu, dk, dv);
      int bits = 1;
      while ((1 << bits) < world)
        bits++;
      rc = sort_pairs_u32(ctx, dk, dks, dv, dvs, n_own, bits);  // (radix sort is stable: the slab order survives inside a destination)
      if (rc != BS_OK)
        return rc;
      int64_t* d_pos = P.buf<int64_t>(SMALL, 64 + world) + 64;
      count_dest_kernel<<<1, 256, 0, st>>>(dks, n_own, world, d_pos);
      std::vector<int64_t> pos(world);
      BS_HIP(ctx, hipMemcpyAsync(pos.data(), d_pos, sizeof(int64_t) * world, hipMemcpyDeviceToHost, st));
      BS_HIP(ctx, hipStreamSynchronize(st));
      for (int r = 0; r < world; r++)
        send[r] = pos[r] - (r ? pos[r - 1] : 0);
    }
    // one all-to-all per array, payload gathered into send order first
    char* sbuf = P.buf<char>(SEND, (size_t)n_own * std::max<size_t>({sizeof(int4), sizeof(int32_t) * (size_t)k, sizeof(double) * 3}) + 64);
    SH_CHECK(sbuf);
    void* got = nullptr;
    if (n_own > 0)
      gather_int4_kernel<<<nblk(n_own, 256), 256, 0, st>>>(own, dvs, n_own, (int4*)sbuf);
    rc = P.all_to_all_rows(sbuf, send, sizeof(int4), G_OWN, &got, &n_loc);
    if (rc != BS_OK)
      return rc;
    g_own = (int4*)got;
    if (n_own > 0)
      gather_rows_kernel<<<nblk(n_own * (int64_t)k, 256), 256, 0, st>>>(neigh, dvs, n_own, k, (int32_t*)sbuf);
    int64_t n2 = 0;
    rc = P.all_to_all_rows(sbuf, send, (int64_t)sizeof(int32_t) * k, G_NG, &got, &n2);
    if (rc != BS_OK)
      return rc;
    g_ng = (int32_t*)got;
    if (n_own > 0)
      gather_rows_kernel<<<nblk(n_own * 6, 256), 256, 0, st>>>((const int32_t*)normals, dvs, n_own, 6, (int32_t*)sbuf);
    rc = P.all_to_all_rows(sbuf, send, (int64_t)sizeof(double) * 3, G_NR, &got, &n2);
    if (rc != BS_OK)
      return rc;
    g_nr = (double*)got;
    lap(inf.ms_redistribute);
  }
  inf.n_grow = n_loc;

  // ---- 6. localize + grow ----
  int32_t* gl = P.buf<int32_t>(VALS_A, 2 * (size_t)n_loc + 32);
  int32_t* io = P.buf<int32_t>(VALS_B, 2 * (size_t)n_loc + 32);
  int32_t* sg = P.buf<int32_t>(SG, (size_t)n_loc + 16);
  SH_CHECK(gl && io && sg);
  int32_t* perm = io + n_loc + 8;
  int32_t* d_bad = (int32_t*)P.buf<int64_t>(SMALL, 64) + 2;
  BS_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
  int lrc = BS_OK;
  int32_t* l_xyz = P.buf<int32_t>(L_XYZ, 3 * (size_t)n_loc + 16);
  int32_t* l_ng = P.buf<int32_t>(L_NG, (size_t)n_loc * k + 16);
  int32_t* l_tmp = P.buf<int32_t>(L_MISC, (size_t)n_loc * std::max(k, 6) + 16);
  double* l_nr = P.buf<double>(L_NR, 3 * (size_t)n_loc + 16);
  SH_CHECK(l_xyz && l_ng && l_tmp && l_nr);
  int64_t tot = n_loc;
  rc = P.all_reduce_host(&tot, 1, BS_SUM);
  if (rc != BS_OK)
    return rc;
  int32_t* labels_l = nullptr;
  int32_t* owner_l = nullptr;
  int32_t np_local = 0;
  if (n_loc > 0) {
    row_gidx_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(g_own, n_loc, gl, io);
    lrc = sort_pairs_i32(ctx, gl, sg, io, perm, n_loc);
    if (lrc == BS_OK) {
      dup_check_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(sg, n_loc, d_bad);
      int4* srt = (int4*)P.buf<char>(SEND, (size_t)n_loc * sizeof(int4) + 64);
      SH_CHECK(srt);
      gather_int4_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(g_own, perm, n_loc, srt);
      unpack_rows_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(srt, n_loc, l_xyz, nullptr);
      gather_rows_kernel<<<nblk(n_loc * 6, 256), 256, 0, st>>>((const int32_t*)g_nr, perm, n_loc, 6, (int32_t*)l_nr);
      gather_rows_kernel<<<nblk(n_loc * (int64_t)k, 256), 256, 0, st>>>(g_ng, perm, n_loc, k, l_tmp);
      int32_t bad = 0;
      if (hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        lrc = fail(ctx, BS_ERR_HIP, "bs_segment_sharded: copy failed");
      else if (bad || tot != n_total)
        lrc = fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: the shards do not cover the cloud exactly once (d_gidx must be a permutation of 0..n_total-1 over all ranks)");
    }
    if (lrc == BS_OK) {
      int32_t miss = 0;
      lrc = bs_remap_rows_dev(ctx, l_tmp, n_loc, k, sg, n_loc, l_ng, &miss);
      if (lrc == BS_OK && miss)
        lrc = fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: a k-list refers to a point outside its connected component");
    }
    if (lrc == BS_OK && n_loc < k)
      lrc = fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: fewer points than k in a shard");
    if (lrc == BS_OK) {
      labels_l = l_tmp;  // (free again: the k-lists are in l_ng)
      bs_params pp = p;
      if (pp.rg_mode == 1)
        pp.rg_mode = 0;  // owners are needed: the speculative grower
      lrc = bs_region_grow_dev(ctx, l_xyz, l_nr, l_ng, n_loc, &pp, labels_l);
    }
    if (lrc == BS_OK) {
      owner_l = labels_l + n_loc + 8 <= l_tmp + (size_t)n_loc * std::max(k, 6) ? labels_l + n_loc + 8 : nullptr;
      if (!owner_l)
        lrc = fail(ctx, BS_ERR_INTERNAL, "bs_segment_sharded: scratch too small");
    }
    if (lrc == BS_OK)
      lrc = bs_owner_fetch_dev(ctx, owner_l);
    if (lrc == BS_OK)
      lrc = bs_plane_seeds_dev(ctx, nullptr, 0, &np_local);
  } else if (tot != n_total) {
    lrc = fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: the shards do not cover the cloud exactly once");
  }
  rc = P.agree(lrc, "bs_segment_sharded: another rank failed in region growing");
  if (rc != BS_OK)
    return rc;
  lap(inf.ms_grow);

  // ---- 7. global plane ids and labels ----
  int64_t npl = np_local;
  std::vector<int64_t> npc;
  rc = P.all_gather_host(&npl, 1, npc);
  if (rc != BS_OK)
    return rc;
  const int64_t capp = std::max<int64_t>(1, *std::max_element(npc.begin(), npc.end()));
  int64_t np_tot = 0;
  for (int64_t v : npc)
    np_tot += v;
  int32_t* sp = P.buf<int32_t>(KEYS_A, (size_t)capp * (world + 2) + (size_t)np_tot * 2 + 64);
  SH_CHECK(sp);
  int32_t* sp_all = sp + capp;
  int32_t* seeds_cat = sp_all + (size_t)capp * world;
  int32_t* seeds_sorted = seeds_cat + np_tot + 8;
  BS_HIP(ctx, hipMemsetAsync(sp, 0, sizeof(int32_t) * capp, st));
  if (np_local > 0) {
    int32_t* sl = P.buf<int32_t>(KEYS_B, (size_t)np_local + 16);
    SH_CHECK(sl);
    int32_t dummy = 0;
    rc = bs_plane_seeds_dev(ctx, sl, np_local, &dummy);
    if (rc != BS_OK)
      return rc;
    lookup_kernel<<<nblk(np_local, 256), 256, 0, st>>>(sg, sl, np_local, sp);  // local seed -> global index
  }
  rc = P.all_gather(sp, sp_all, (int64_t)sizeof(int32_t) * capp);
  if (rc != BS_OK)
    return rc;
  {
    int64_t off = 0;
    for (int r = 0; r < world; r++) {
      if (npc[r] > 0)
        BS_HIP(ctx, hipMemcpyAsync(seeds_cat + off, sp_all + (size_t)r * capp, sizeof(int32_t) * npc[r], hipMemcpyDeviceToDevice, st));
      off += npc[r];
    }
  }
  if (np_tot > 0) {
    rc = sort_keys<int32_t>(ctx, seeds_cat, seeds_sorted, np_tot);
    if (rc != BS_OK)
      return rc;
  }
  fill_kernel<<<nblk(n_total, 256), 256, 0, st>>>(d_plane_idx, n_total, -1);
  if (n_loc > 0) {
    int32_t* owner_g = P.buf<int32_t>(KEYS_B, 2 * (size_t)n_loc + 32);
    SH_CHECK(owner_g);
    int32_t* lab_g = owner_g + n_loc + 8;
    lookup_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(sg, owner_l, n_loc, owner_g);
    rc = bs_labels_from_owner_dev(ctx, owner_g, n_loc, seeds_sorted, (int32_t)np_tot, lab_g);
    if (rc != BS_OK)
      return rc;
    scatter_labels_kernel<<<nblk(n_loc, 256), 256, 0, st>>>(sg, lab_g, n_loc, d_plane_idx);
  }
  rc = P.all_reduce(d_plane_idx, n_total, BS_I32, BS_MAX);  // every point has exactly one growing rank; the others hold -1
  if (rc != BS_OK)
    return rc;
  inf.planes_total = np_tot;
  // what bs_sharded_planes_fetch needs: global indices of the local points, all committed seeds
  ctx->sh_seeds.resize((size_t)np_tot);
  if (np_tot > 0)
    BS_HIP(ctx, hipMemcpyAsync(ctx->sh_seeds.data(), seeds_sorted, sizeof(int32_t) * np_tot, hipMemcpyDeviceToHost, st));
  BS_HIP(ctx, hipStreamSynchronize(st));
  BS_HIP(ctx, hipGetLastError());
  ctx->sh_nloc = n_loc;
  ctx->sh_valid = n_loc >= k;
  lap(inf.ms_labels);
  if (info)
    *info = inf;
  return BS_OK;
}

}  // namespace

}  // namespace bs

using namespace bs;

extern "C" {

int bs_comm_rccl(void* nccl_comm, int32_t rank, int32_t world, bs_comm_ops* out)
{
  if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world)
    return BS_ERR_INVALID;
  if (!rccl().ok)
    return BS_ERR_NO_DEVICE;
  auto* h = new (std::nothrow) RcclHandle{(ncclComm_t)nccl_comm, rank, world};
  if (!h)
    return BS_ERR_NOMEM;
  out->handle = h;  // (a few bytes per communicator, released with the process)
  out->rank = rank;
  out->world = world;
  out->all_reduce = rccl_all_reduce;
  out->all_gather = rccl_all_gather;
  out->all_to_all_v = rccl_all_to_all_v;
  return BS_OK;
}

int bs_comm_rccl_unique_id(char id[128])
{
  if (!id)
    return BS_ERR_INVALID;
  if (!rccl().ok)
    return BS_ERR_NO_DEVICE;
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess)
    return BS_ERR_INTERNAL;
  static_assert(sizeof u.internal == 128, "ncclUniqueId is 128 bytes");
  memcpy(id, u.internal, 128);
  return BS_OK;
}

int bs_comm_rccl_init(bs_ctx* ctx, const char id[128], int32_t rank, int32_t world, void** nccl_comm)
{
  if (!ctx || !id || !nccl_comm || world < 1 || rank < 0 || rank >= world)
    return BS_ERR_INVALID;
  if (!rccl().ok)
    return fail(ctx, BS_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(u.internal, id, 128);
  ncclComm_t c = nullptr;
  if (rccl().CommInitRank(&c, world, u, rank) != ncclSuccess)
    return fail(ctx, BS_ERR_INTERNAL, "ncclCommInitRank failed");
  *nccl_comm = c;
  return BS_OK;
}

int bs_comm_rccl_destroy(void* nccl_comm)
{
  if (!nccl_comm)
    return BS_ERR_INVALID;
  if (!rccl().ok)
    return BS_ERR_NO_DEVICE;
  return rccl().CommDestroy((ncclComm_t)nccl_comm) == ncclSuccess ? BS_OK : BS_ERR_INTERNAL;
}

int bs_comm_local_create(int32_t world, bs_comm_ops* out /* [world] */)
{
  if (world < 1 || !out)
    return BS_ERR_INVALID;
  auto* S = new (std::nothrow) LocalShared(world);
  if (!S)
    return BS_ERR_NOMEM;
  for (int r = 0; r < world; r++) {
    out[r].handle = new LocalHandle{S, r};
    out[r].rank = r;
    out[r].world = world;
    out[r].all_reduce = local_all_reduce;
    out[r].all_gather = local_all_gather;
    out[r].all_to_all_v = local_all_to_all_v;
  }
  return BS_OK;
}

void bs_comm_local_destroy(bs_comm_ops* ops)
{
  if (!ops || !ops->handle)
    return;
  auto* h = (LocalHandle*)ops->handle;
  LocalShared* S = h->sh;
  delete h;
  ops->handle = nullptr;
  if (S->refs.fetch_sub(1) == 1)
    delete S;
}

int bs_segment_sharded(bs_ctx* ctx, const bs_comm_ops* comm, const int32_t* d_xyz, const int32_t* d_gidx, int64_t m,
                       int64_t n_total, const bs_params* p, double halo, int32_t* d_plane_idx, bs_shard_info* info)
{
  if (!ctx)
    return BS_ERR_INVALID;
  if (!p || !d_plane_idx || m < 0 || (m > 0 && !d_xyz) || n_total <= 0 || n_total >= (int64_t)INT32_MAX - 64 || m > n_total)
    return fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: bad arguments");
  if (comm && (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || !comm->all_reduce || !comm->all_gather ||
               !comm->all_to_all_v))
    return fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: incomplete bs_comm_ops");
  if (comm && comm->world > 1 && !d_gidx && m > 0)
    return fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: d_gidx is required when the cloud is spread over several ranks");
  if (p->k < 2 || p->k > 32 || n_total < p->k)
    return fail(ctx, BS_ERR_INVALID, "bs_segment_sharded: k out of range or n_total < k");
  BS_HIP(ctx, hipSetDevice(ctx->device));
  return sharded_impl(ctx, comm, d_xyz, d_gidx, m, n_total, *p, halo, d_plane_idx, info);
}

int bs_sharded_planes_fetch(bs_ctx* ctx, bs_planes* out)
{
  if (!ctx || !out)
    return BS_ERR_INVALID;
  memset(out, 0, sizeof *out);
  if (!ctx->sh_valid) {  // this rank grew nothing: an empty set
    out->offset = (int64_t*)calloc(1, sizeof(int64_t));
    return out->offset ? BS_OK : BS_ERR_NOMEM;
  }
  int rc = bs_planes_fetch(ctx, out);
  if (rc != BS_OK)
    return rc;
  // global index of every local point (the sorted global indices are still in the pass's scratch)
  std::vector<int32_t> sg((size_t)ctx->sh_nloc);
  if (hipMemcpy(sg.data(), ctx->sh[SG].p, sizeof(int32_t) * sg.size(), hipMemcpyDeviceToHost) != hipSuccess) {
    bs_planes_free(out);
    return fail(ctx, BS_ERR_HIP, "bs_sharded_planes_fetch: copy failed");
  }
  const int64_t total = out->offset[out->n_planes];
  for (int64_t t = 0; t < total; t++)
    out->point_idx[t] = sg[(size_t)out->point_idx[t]];
  for (int i = 0; i < out->n_planes; i++) {
    const int32_t seed = out->point_idx[out->offset[i]];  // pointIdx[0] is the seed (my_function.cpp:191)
    out->id[i] = 1 + (int32_t)(std::lower_bound(ctx->sh_seeds.begin(), ctx->sh_seeds.end(), seed) - ctx->sh_seeds.begin());
  }
  return BS_OK;
}

}  // extern "C"
