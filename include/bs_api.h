/*
 * bs_api.h -- C ABI of the MI355X-native buildingSegment hot path
 * (kNN neighbourhood build -> PCA normal estimation -> region-growing plane
 * labelling).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Reference interfaces this boundary replaces (all paths relative to
 * /root/reference/):
 *   tmc3/TMC3.cpp:213-218        the four call-site lines in main()
 *   tmc3/my_function.h:48-85     get_Normal_and_K_neighbor<K>()      -> bs_knn_normals*
 *   tmc3/my_function.h:89-123    class seg_plane (ctor, get_planes)  -> bs_region_grow*
 *   tmc3/my_function.cpp:180-258 seg_plane::get_planes / Broad       -> bs_region_grow*
 *   tmc3/my_function.h:25-30     struct plane                        -> bs_planes (CSR)
 *   tmc3/my_function.cpp:260-275 seg_plane::set_plane_color          -> bs_plane_colors
 *   tmc3/PCCPointSet.h:60,67,605 positions (int32 AoS) / planeIdx    -> xyz / plane_idx buffers
 *
 * Conventions
 *   - xyz is the reference's AoS layout: int32 [n][3] millimetres == &cloud[0].
 *   - neigh is row-major int32 [n][k], nearest first, ties (equal squared
 *     distance) broken by ascending point index, self included.
 *   - normals are f64 [n][3], unit length, oriented to +z.
 *   - plane_idx is int32 [n]: -1 = unlabelled, >=1 = plane id (orphans of
 *     failed seeds carry the id of the next committed plane, exactly as the
 *     reference leaves them).
 *   - every function returns BS_OK (0) or a negative bs_status; nothing aborts.
 *   - *_dev entry points take DEVICE pointers, enqueue on the context's
 *     stream and do not synchronise unless stated.
 *   - a bs_ctx is owned by one host thread at a time.
 */
#ifndef BS_API_H
#define BS_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BS_API_VERSION 5

typedef enum bs_status {
  BS_OK = 0,
  BS_ERR_INVALID = -1,     /* null pointer, n < k, k out of range, bad params */
  BS_ERR_RANGE = -2,       /* coordinates outside the exactly-representable domain */
  BS_ERR_NOMEM = -3,       /* host or device allocation failed */
  BS_ERR_HIP = -4,         /* HIP runtime error (see bs_last_error) */
  BS_ERR_NO_DEVICE = -5,   /* no usable gfx950 device */
  BS_ERR_INTERNAL = -6,    /* invariant violated (watchdog, overflow of a work list) */
  BS_ERR_UNCERTIFIED = -7  /* halo too thin: some queries could not be certified */
} bs_status;

/* Parameters.  Defaults are the reference's literals:
 *   k 15 (TMC3.cpp:215-216), radius 100 / max_nn 50 (my_function.h:63),
 *   th_thickness 300, th_point_count 400 (my_function.h:117-118),
 *   cos_th 0.88 (my_function.cpp:230). */
typedef struct bs_params {
  int32_t k;              /* neighbour-list length, 2..32 */
  int32_t max_nn;         /* hybrid search cap for normals, 3..64 */
  double radius;          /* hybrid search radius (mm), strict d^2 < radius^2 */
  int32_t th_thickness;   /* region grow: |distance to plane| <= th_thickness */
  int32_t th_point_count; /* keep a plane iff pointIdx.size() > th_point_count */
  double cos_th;          /* region grow: cur_normal . normal[id] >= cos_th */
  int32_t cell_size;      /* search-grid cell edge in mm, 0 = choose automatically */
  int32_t rg_mode;        /* 0 = auto, 1 = single-wave sequential, 2 = multi-plane speculative */
} bs_params;

/* CSR form of std::vector<plane> (my_function.h:25-30), in the reference's
 * commit order.  point_idx keeps the reference's list order and duplicates
 * (the seed can appear twice).  Arrays are owned by the library after a
 * successful call and released by bs_planes_free. */
typedef struct bs_planes {
  int32_t n_planes;
  int32_t* id;        /* [n_planes]   plane::id (1-based)            */
  double* normal;     /* [n_planes*3] plane::normal                  */
  int32_t* center;    /* [n_planes*3] plane::center                  */
  int64_t* offset;    /* [n_planes+1] CSR offsets into point_idx     */
  int32_t* point_idx; /* [offset[n_planes]] plane::pointIdx, in order */
} bs_planes;

/* Per-call stage timings (milliseconds of device time, HIP events on the
 * context's stream) of the last *_dev / host call. */
typedef struct bs_timings {
  double grid_ms;    /* cell keys + sort + cell table                */
  double knn_ms;     /* kNN + normals kernels (incl. fallback rings) */
  double grow_ms;    /* region growing                               */
  double total_ms;   /* first kernel to last kernel                  */
  int64_t largest_plane;  /* pointIdx.size() of the largest committed plane */
  int64_t n_seed_attempts;
  int64_t n_fallback_queries; /* queries that left the LDS-tile fast path */
  int64_t rg_rounds;      /* speculative rounds (rg_mode 2) */
  double grow_kernel_ms;  /* sum of the plane-growth kernel launches alone (HIP events around each launch) */
  int64_t grow_kernel_launches;
  double grow_setup_ms;   /* region growing, one-time part: records, static masks, reverse lists, first owner fixed point */
  int64_t validation_rejects; /* finished planes (or duplicated list entries) the post-round validation refused;
                                 refused planes are grown again, the result stays exact */
  int64_t forged_seed;        /* seed of the plane bs_selftest_forge_next corrupted, -1 if none */
  int64_t forged_refused;     /* 1 if the validation refused exactly that plane */
  int64_t audit_attempts;     /* bs_set_audit: plane attempts replayed against the final owners (-1: audit off) */
  int64_t audit_mismatches;   /* ... of which differ from what was committed (must be 0) */
  double audit_ms;            /* time of the replay + comparison (not part of grow_ms / total_ms) */
  /* -- API version 5 -- */
  int64_t tie_rows;           /* queries of the last kNN call whose k-list has an equal-d^2 pair inside it or at its
                                 boundary (k-th vs (k+1)-th neighbour): the only rows where the reference's kd-tree
                                 traversal order may differ from this library's canonical (d^2, index) order */
  /* why the post-round validation refused finished planes (every refused plane is grown again; the result stays
   * exact).  validate1: the plane was robbed after it finished / a list entry no longer carries its claim / a point
   * is listed twice; validate3: normal or centre not reproducible from the list. */
  int64_t rej_v1_robbed, rej_v1_tag, rej_v1_dup, rej_v3_state;
  /* finished planes that were merely INCONSISTENT with the settled owners (normal outcome of the speculation, not
   * a refusal): seed row no longer free / an accepted point was taken earlier / a logged assumption failed */
  int64_t incons_seed, incons_list, incons_log;
} bs_timings;

/* sizeof(bs_timings) of the library build: a host compiled against another header version must not call
 * bs_get_timings (the struct is written whole).  host/bs_legacy.hpp and the Python loader check it. */
int64_t bs_sizeof_timings(void);

typedef struct bs_ctx bs_ctx;

int bs_api_version(void);
const char* bs_strerror(int status);
void bs_params_default(bs_params* p);

/* Context: binds one HIP device; owns the stream and all scratch buffers. */
int bs_create(int device, bs_ctx** out);
void bs_destroy(bs_ctx* ctx);
const char* bs_last_error(const bs_ctx* ctx);
/* Use an existing hipStream_t (e.g. the caller's current stream); NULL
 * restores the context's own stream. */
int bs_set_stream(bs_ctx* ctx, void* hip_stream);
int bs_get_timings(const bs_ctx* ctx, bs_timings* out);

/* ---- host-buffer entry points (what the reference's call sites bind) ---- */

/* Replaces get_Normal_and_K_neighbor<K>() (my_function.h:48-85). */
int bs_knn_normals(bs_ctx* ctx, const int32_t* xyz, int64_t n, const bs_params* p,
                   int32_t* neigh /* [n][k] out */, double* normals /* [n][3] out */);

/* Replaces seg_plane::seg_plane + get_planes (my_function.h:98-104,
 * my_function.cpp:180-258). */
int bs_region_grow(bs_ctx* ctx, const int32_t* xyz, const double* normals,
                   const int32_t* neigh, int64_t n, const bs_params* p,
                   int32_t* plane_idx /* [n] out */, bs_planes* planes /* out, may be NULL */);

/* Whole path, device-resident between the stages (TMC3.cpp:213-217). */
int bs_segment(bs_ctx* ctx, const int32_t* xyz, int64_t n, const bs_params* p,
               int32_t* neigh /* may be NULL */, double* normals /* may be NULL */,
               int32_t* plane_idx, bs_planes* planes /* may be NULL */);

/* Halo (multi-GPU slab) form of bs_knn_normals: the first n_query points of a
 * local cloud are the queries, the rest is halo; gidx [n] gives every local
 * point's global index (ties are broken by it and neigh holds global
 * indices).  A k-list is certified iff its k-th distance < cert_radius (the
 * halo width); *n_uncertified counts the failures (results are still written). */
int bs_knn_normals_halo(bs_ctx* ctx, const int32_t* xyz, const int32_t* gidx, int64_t n, int64_t n_query,
                        const bs_params* p, int32_t* neigh /* [n_query][k] */, double* normals /* [n_query][3] */,
                        double cert_radius, int64_t* n_uncertified);

void bs_planes_free(bs_planes* planes);

/* Replaces seg_plane::set_plane_color (my_function.cpp:260-275): colours
 * [n][3] uint16 in the reference's internal G,B,R slot order, all zero, then
 * one colour per plane applied to every pointIdx entry.  plane_rgb is
 * [n_planes][3], the values the caller drew (the reference draws
 * 55 + rand() % 200 three times per plane). */
int bs_plane_colors(const bs_planes* planes, const int32_t* plane_rgb, int64_t n,
                    uint16_t* colors /* [n][3] out */);

/* ---- device-buffer entry points (the measured path) ---- */

/* kNN + normals for the queries [q_begin, q_end) of a device-resident cloud of
 * n points.  d_gidx (nullable) gives the global index of each local point
 * (used for tie-breaking and written into d_neigh); NULL = identity.
 * Outputs are indexed by (query - q_begin).  cert_radius > 0 certifies a
 * k-list only if its k-th distance is < cert_radius (halo mode);
 * *n_uncertified (host, nullable) receives the number of failures after a
 * stream sync. */
int bs_knn_normals_dev(bs_ctx* ctx, const int32_t* d_xyz, const int32_t* d_gidx, int64_t n,
                       int64_t q_begin, int64_t q_end, const bs_params* p,
                       int32_t* d_neigh, double* d_normals, double cert_radius,
                       int64_t* n_uncertified);

/* Region growing on device-resident inputs; d_plane_idx [n] out.  Plane
 * records stay on the device until bs_planes_fetch. Synchronises the stream
 * (the speculative scheduler is host-driven). */
int bs_region_grow_dev(bs_ctx* ctx, const int32_t* d_xyz, const double* d_normals,
                       const int32_t* d_neigh, int64_t n, const bs_params* p,
                       int32_t* d_plane_idx);

/* Fused device-resident pipeline: xyz in HBM -> plane_idx in HBM.
 * d_neigh / d_normals may be NULL (scratch owned by the context is used). */
int bs_segment_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, const bs_params* p,
                   int32_t* d_neigh, double* d_normals, int32_t* d_plane_idx);

/* Pre-/post-processing either side of the path, on the device (SURVEY.md 8f-2,3).
 *
 * bs_shift_to_origin_dev: the buildingSeg constructor's bounding-box shift
 * (TMC3.cpp:55-73): min over the cloud, then xyz -= min IN PLACE; min_out [3]
 * (host, nullable) receives the subtracted minimum.  Synchronises.
 *
 * bs_plane_colors_dev: seg_plane::set_plane_color (my_function.cpp:260-275) for
 * the planes of the last region grow on this context: d_colors [n][3] uint16
 * (G,B,R slots) zeroed, then plane_rgb[p] (host, [n_planes][3]) scattered to
 * every pointIdx entry of plane p. */
int bs_shift_to_origin_dev(bs_ctx* ctx, int32_t* d_xyz, int64_t n, int32_t* min_out);

/* PLY ingest on the device (replaces the per-property loop of
 * ply::read, ply.cpp:431-501, for the positions): d_records is the binary
 * little-endian vertex body resident on the device (n records of `stride` bytes,
 * x/y/z at byte offsets off_x/off_y/off_z, all three float32 (is_f64 = 0) or
 * float64).  d_xyz [n][3] receives (int32) trunc(value * scale) (ply.cpp:436-465;
 * float promoted to double first).  shift_to_origin != 0 additionally applies the
 * buildingSeg constructor's bounding-box shift (TMC3.cpp:55-73) and reports the
 * subtracted minimum in min_out (host [3], nullable).  BS_ERR_RANGE if a product
 * does not fit int32 (undefined behaviour in the reference).  Synchronises. */
int bs_ingest_dev(bs_ctx* ctx, const void* d_records, int64_t n, int32_t stride, int32_t off_x, int32_t off_y,
                  int32_t off_z, int32_t is_f64, double scale, int32_t shift_to_origin, int32_t* d_xyz,
                  int32_t* min_out);
int bs_plane_colors_dev(bs_ctx* ctx, const int32_t* plane_rgb, int32_t n_planes, int64_t n, uint16_t* d_colors);

/* 2-D density / height raster: the reference's (currently commented-out) 2-D
 * branch, SURVEY.md 8f-4.  Replaces buildingSeg::groundTH (TMC3.cpp:183-199) and
 * buildingSeg::compute_gird_picture (TMC3.cpp:123-174) for a cloud that has
 * already been shifted to its bounding-box origin (bs_shift_to_origin_dev, i.e.
 * the buildingSeg constructor, TMC3.cpp:55-73).
 *
 *   extent[3]  = box.max - box.min of the unshifted cloud (= max of the shifted one)
 *   bin        = pixel edge in mm (reference: 100), bin_height = height-histogram
 *                bin in mm (reference: 1000)
 *   bs_grid_dims: width = extent[0]/bin + 2, height = extent[1]/bin + 2 (TMC3.cpp:75-76)
 *   image      = [height][width][3] f64, caller-allocated, pixel(x,y,c) at
 *                (y*width + x)*3 + c as in TMC3.cpp:119-121:
 *                  c=0 mean height of the splatted points, c=1 log(density+1) (+20
 *                  where non-zero), c=2 zero (the reference never writes it)
 *   ground_th  (host, nullable) receives groundTH()
 * Bit-identical to the reference's sequential f64 accumulation in channel 0 and in
 * the density sums; the logarithm is bs_det_log of include/bs_detmath.h (within
 * 1 ulp of the platform libm the reference calls).  BS_ERR_RANGE if a coordinate
 * lies outside [0, extent].  Synchronises. */
int bs_grid_dims(const int32_t* extent, int32_t bin, int32_t* width, int32_t* height);
int bs_grid_picture(bs_ctx* ctx, const int32_t* xyz, int64_t n, const int32_t* extent, int32_t bin,
                    int32_t bin_height, double* image, double* ground_th);
int bs_grid_picture_dev(bs_ctx* ctx, const int32_t* d_xyz, int64_t n, const int32_t* extent, int32_t bin,
                        int32_t bin_height, double* d_image, double* ground_th);

/* Self-test of the grower's plane-centre division (csrc/bs_centerdiv.h) ON THE
 * DEVICE: out[i] = (int32_t)((uint64_t)(int64_t)c[i] / n[i]), the expression of
 * my_function.cpp:249-250, for count host-side pairs (1 <= n[i] < 2^31).  The
 * device version rests on the hardware reciprocal seed, which no host test can
 * exercise; tests/test_gpu_parity.py checks millions of pairs through this. */
int bs_selftest_center_div(bs_ctx* ctx, const int32_t* c, const uint32_t* n, int32_t* out, int64_t count);

/* Self-test of the grower's post-round validation: the NEXT region grow on this context corrupts one
 * finished plane before validating it -- mode 1: a list entry duplicated (a point held twice), mode 2:
 * the reported normal off by one ulp -- the way the claim-protocol bugs found by fuzzing did.  The
 * validation must refuse the plane (bs_timings.forged_refused == 1, forged_seed names it); it is
 * then grown again and the final result is still exact. */
int bs_selftest_forge_next(bs_ctx* ctx, int mode);

/* Audit mode (off by default; also switched on by the environment variable BS_AUDIT=1).  After a
 * speculative region grow (rg_mode 0 / 2) every plane attempt that exists under the FINAL owners -- the
 * committed planes and the attempts the reference rolls back (my_function.cpp:199) -- is grown once more,
 * all of them concurrently, with the production step engine but WITHOUT speculation: a point is taken iff
 * its final owner is an earlier attempt, nothing is assumed, nothing can be stolen.  Every Broad() decision
 * of every plane is thereby re-made against the state the sequential reference has at that plane's time;
 * the replayed lists must equal the committed ones entry by entry, in order, with bit-equal normal and
 * centre, and an attempt that was not committed must end at or below the commit threshold.  Together with
 * the owner equations (BS_VERIFY=1) this certifies the result of the speculation by induction over the
 * seed index.  Costs about one extra pass of the longest plane; bs_timings.audit_* report it. */
int bs_set_audit(bs_ctx* ctx, int on);

/* Copy the plane records of the last region-grow on this context to the host. */
int bs_planes_fetch(bs_ctx* ctx, bs_planes* planes);

/* ---- building blocks of the multi-GPU path (component-sharded stage 3) ----
 *
 * The reference's seed scan (my_function.cpp:184-217) is global, but information only travels along kNN edges
 * (Broad tests neigh[Idx][1..K-1], :224-233; a failed seed labels part of its own row, :238-239): connected
 * components of the kNN graph never interact, and the only shared state is cur_planeId, which advances once per
 * committed plane (:199-202), i.e.
 *     planeIdx[p] = 1 + #(committed seeds < owner[p]),   owner[p] = the seed attempt that left p labelled.
 * So stage 3 shards exactly: whole components per GPU (local cloud in ascending global index order), committed
 * seeds all-gathered, labels from the owners.  buildingsegment_amd/dist.py (torch.distributed) and
 * bs_segment_sharded (RCCL, below) are built from these calls.
 *
 * bs_cc_hook_dev: one hooking step of a distributed union-find.  d_parent [n_total] (device, in/out) is a parent
 *   array over GLOBAL point ids with parent[x] <= x (start: identity).  The call unites u = d_gidx[i] (NULL: i) with
 *   every d_rows[i][j] (global ids, [m][k]) and points every node it touched straight at its root.  *n_hooks (host)
 *   = unions performed.  Multi-GPU: every rank hooks its own rows, the parent arrays are all-reduced with MIN, and
 *   the step repeats until no rank hooked anything; then parent[x] = smallest global id of x's component.
 *   Synchronises.
 * bs_owner_fetch_dev: d_owner [n] (device) receives, in the caller's index order, the seed index of the attempt
 *   that left each point labelled after the last bs_region_grow[_dev] / bs_segment[_dev] on this context (-1:
 *   unlabelled).  rg_mode 0 / 2 only (the single-wave grower keeps no owners: BS_ERR_INVALID).
 * bs_labels_from_owner_dev: d_plane_idx[i] = d_owner[i] < 0 ? -1 : 1 + #(d_seeds[] < d_owner[i]); d_seeds is the
 *   ascending list of ALL committed seeds (global indices), d_owner holds global seed indices.
 * bs_remap_rows_dev: d_out[t] = position of d_rows[t] in the ascending array d_sorted_gidx [n] (global -> local
 *   index of a component-complete local cloud); *n_missing (host) != 0 if some index was not found.  Synchronises. */
int bs_cc_hook_dev(bs_ctx* ctx, const int32_t* d_rows, const int32_t* d_gidx, int64_t m, int32_t k, int32_t* d_parent,
                   int64_t n_total, int64_t* n_hooks);
int bs_owner_fetch_dev(bs_ctx* ctx, int32_t* d_owner);
/* bs_plane_seeds_dev: the seeds (= pointIdx[0], my_function.cpp:191) of the planes the last speculative grow on this
 *   context committed, ascending (= commit order); *n_planes (host) receives their number, at most cap of them are
 *   copied to d_seeds (device, nullable).  bs_stream_sync: wait for everything enqueued on the context's stream. */
int bs_plane_seeds_dev(bs_ctx* ctx, int32_t* d_seeds, int64_t cap, int32_t* n_planes);
int bs_stream_sync(bs_ctx* ctx);
int bs_labels_from_owner_dev(bs_ctx* ctx, const int32_t* d_owner, int64_t n, const int32_t* d_seeds, int32_t n_seeds,
                             int32_t* d_plane_idx);
int bs_remap_rows_dev(bs_ctx* ctx, const int32_t* d_rows, int64_t n_rows, int32_t k, const int32_t* d_sorted_gidx,
                      int64_t n, int32_t* d_out, int32_t* n_missing);

/* ---- multi-GPU entry point: ONE cloud sharded over the ranks of a communicator ----
 *
 * bs_segment_sharded is what the C++ host of TMC3.cpp:213-217 calls when it runs as one process per GPU: rank r
 * passes the points it holds (any split of the cloud: d_xyz [m][3], d_gidx [m] = their global indices, both on
 * its device) and receives planeIdx for the WHOLE cloud (d_plane_idx [n_total], device, identical on every rank).
 * Stages 1-2 run on Morton slabs with a halo exchange, stage 3 on whole connected components of the kNN graph
 * (see "building blocks" above); results equal the single-GPU / sequential ones bit for bit.
 *
 * The collectives are reached through bs_comm_ops: three operations on DEVICE buffers, enqueued on `stream`
 * (the context's stream; they may synchronise it).  bs_comm_rccl() fills the table for an RCCL communicator
 * (ncclComm_t; librccl.so.1 is resolved at run time) -- reductions by ncclAllReduce, gathers by ncclAllGather,
 * the split-size all-to-all by grouped ncclSend / ncclRecv over xGMI.  A host with another transport (MPI, a test
 * harness) supplies its own three functions.  Every function returns 0 or a non-zero error. */
typedef enum bs_comm_dtype { BS_I32 = 0, BS_I64 = 1 } bs_comm_dtype;
typedef enum bs_comm_op { BS_MIN = 0, BS_MAX = 1, BS_SUM = 2 } bs_comm_op;
typedef struct bs_comm_ops {
  void* handle; /* passed back as the first argument */
  int32_t rank, world;
  /* in-place all-reduce of count elements */
  int (*all_reduce)(void* handle, void* d_buf, int64_t count, int dtype, int op, void* stream);
  /* every rank contributes `bytes` bytes; d_recv receives world * bytes in rank order */
  int (*all_gather)(void* handle, const void* d_send, void* d_recv, int64_t bytes, void* stream);
  /* d_send holds the blocks for ranks 0..world-1 back to back (send_bytes[r] each, host array); d_recv receives the
   * blocks of ranks 0..world-1 back to back (recv_bytes[r] each, host array, already agreed on by the caller) */
  int (*all_to_all_v)(void* handle, const void* d_send, const int64_t* send_bytes, void* d_recv,
                      const int64_t* recv_bytes, void* stream);
} bs_comm_ops;

/* RCCL-backed table for an existing communicator (`nccl_comm` is an ncclComm_t created by the host for this
 * device).  BS_ERR_NO_DEVICE if librccl cannot be loaded. */
int bs_comm_rccl(void* nccl_comm, int32_t rank, int32_t world, bs_comm_ops* out);
/* Convenience for hosts without RCCL code of their own: rank 0 draws a 128-byte id and hands it to the other
 * ranks by any means; every rank then creates its communicator for the context's device. */
int bs_comm_rccl_unique_id(char id[128]);
int bs_comm_rccl_init(bs_ctx* ctx, const char id[128], int32_t rank, int32_t world, void** nccl_comm);
int bs_comm_rccl_destroy(void* nccl_comm);

/* In-process communicator: the `world` ranks are THREADS of one process (one host thread per GPU without a
 * launcher; several ranks sharing one GPU in the tests).  Fills out[0..world-1]; every rank's buffers must be
 * reachable from the others' devices (same device, or peer access enabled by the host).  Collectives are device
 * copies between the ranks' buffers behind a host barrier -- functional, not tuned: use bs_comm_rccl across GPUs. */
int bs_comm_local_create(int32_t world, bs_comm_ops* out /* [world] */);
void bs_comm_local_destroy(bs_comm_ops* ops /* one entry of the array; call for every rank */);

typedef struct bs_shard_info {
  int64_t n_own;        /* points of this rank's Morton slab */
  int64_t n_local;      /* slab + halo */
  int64_t n_grow;       /* points of the components this rank grew */
  int64_t components;   /* connected components of the kNN graph (whole cloud) */
  int64_t planes_total; /* committed planes (whole cloud) */
  int32_t cc_iterations, halo_retries;
  double halo_mm;
  double ms_partition, ms_halo, ms_knn, ms_components, ms_redistribute, ms_grow, ms_labels; /* host wall time */
} bs_shard_info;

/* halo: initial halo width in mm (0 = 2 * radius); doubled until every k-list is certified.  d_gidx may be NULL
 * only at world 1 (identity).  info is optional. */
int bs_segment_sharded(bs_ctx* ctx, const bs_comm_ops* comm, const int32_t* d_xyz, const int32_t* d_gidx, int64_t m,
                       int64_t n_total, const bs_params* p, double halo, int32_t* d_plane_idx, bs_shard_info* info);
/* The planes THIS rank grew in the last bs_segment_sharded, with global ids (1-based rank of the seed among all
 * committed seeds) and global point indices, ascending id.  Released by bs_planes_free. */
int bs_sharded_planes_fetch(bs_ctx* ctx, bs_planes* planes);

#ifdef __cplusplus
}
#endif
#endif /* BS_API_H */
