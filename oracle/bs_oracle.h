/*
 * bs_oracle.h -- CPU restatement (plain C, single thread) of the
 * buildingSegment hot path.  THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (buildingsegment_amd/csrc) never links or calls it.
 *
 * Pinning status
 *   stage 3 (region grow)  : PINNED -- bit-identical to the reference's own
 *                            seg_plane code compiled verbatim from
 *                            /root/reference (oracle/ref/build_ref.sh ->
 *                            oracle/_ref/ref_stage3) on the fixtures under
 *                            tests/golden/ (tests/test_oracle_golden.py).
 *   stages 1-2 (kNN, normal): PARITY UNPINNED at the Open3D boundary -- the
 *                            arithmetic lives in Open3D 0.19.0 (nanoflann +
 *                            Eigen), which is not vendored in the reference
 *                            and is absent here; the reference holds no
 *                            golden vectors.  This file restates the
 *                            published algorithms and
 *                            is cross-checked against scipy.spatial.cKDTree
 *                            and numpy.linalg.eigh (tests/test_oracle_knn.py).
 */
#ifndef BS_ORACLE_H
#define BS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bso_planes {
  int32_t n_planes;
  int32_t* id;        /* [n_planes] */
  double* normal;     /* [n_planes*3] */
  int32_t* center;    /* [n_planes*3] */
  int64_t* offset;    /* [n_planes+1] */
  int32_t* point_idx; /* [offset[n_planes]] */
} bso_planes;

/* Exact k nearest neighbours (self included), ascending squared distance,
 * ties by ascending index; restates KDTreeFlann::SearchKNN as called at
 * /root/reference/tmc3/my_function.h:71-78 with the canonical tie order.
 * Queries [q0, q1); neigh is [(q1-q0)][k].  cell = grid cell edge (0 = auto).
 * normals (nullable) [(q1-q0)][3]: EstimateNormals(Hybrid(radius, max_nn)) +
 * OrientNormalsToAlignWithDirection((0,0,1)), my_function.h:63-64. */
int bso_knn_normals(const int32_t* xyz, int64_t n, int64_t q0, int64_t q1, int k, double radius,
                    int max_nn, int cell, int32_t* neigh, double* normals);

/* O(n^2) brute-force kNN with the same canonical order (cross-check only). */
int bso_knn_brute(const int32_t* xyz, int64_t n, int64_t q0, int64_t q1, int k, int32_t* neigh);

/* Normal from an explicit neighbour index list (Appendix A.2 + A.3 + orient). */
void bso_normal_from_list(const int32_t* xyz, const int32_t* idx, int cnt, double out[3]);

/* Smallest eigenvector of a symmetric 3x3 (Open3D FastEigen3x3 restated).
 * c = {c00, c01, c02, c11, c12, c22}. */
void bso_fast_eigen3x3(const double c[6], double out[3]);

/* Region growing: restates seg_plane::get_planes / Broad
 * (/root/reference/tmc3/my_function.cpp:180-258) with running sums and an
 * explicit stack; quirks Q1-Q7 reproduced. */
int bso_region_grow(const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                    int k, int th_thickness, int th_point_count, double cos_th,
                    int32_t* plane_idx, bso_planes* planes, int64_t* n_seed_attempts);

/* The same, additionally reporting owner[p] = index of the seed attempt that left p labelled (-1: none):
 * plane_idx[p] == 1 + #(committed seeds < owner[p]).  Checker of bs_owner_fetch_dev and of the
 * component-sharded stage 3 (buildingsegment_amd/dist.py). */
int bso_region_grow_owner(const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                          int k, int th_thickness, int th_point_count, double cos_th,
                          int32_t* plane_idx, bso_planes* planes, int64_t* n_seed_attempts, int32_t* owner);

void bso_planes_free(bso_planes* planes);

double bso_det_acos(double x);
double bso_det_cos(double x);
double bso_det_log(double x);

/*
 * 2-D density / height raster of the reference's (currently disabled) 2-D branch:
 * buildingSeg::buildingSeg dims (TMC3.cpp:75-76), groundTH (:183-199) and
 * compute_gird_picture (:123-174).  xyz is the cloud AFTER the constructor's
 * bounding-box shift (all coordinates >= 0); extent = box.max - box.min.
 *   width = extent[0]/bin + 2, height = extent[1]/bin + 2,
 *   image[(y*width + x)*3 + c]: c=0 mean height above nothing (sum(s*z)/sum(s)),
 *   c=1 log(sum(s)+1) (+20 where non-zero), c=2 untouched (0).
 * Every point at or above the ground threshold (the 1000-mm height bin in which
 * the running count first exceeds n/2) is splatted bilinearly into 4 pixels, in
 * point order -- the f64 sums depend on that order.
 * libm_log != 0: std::log as the reference; 0: bs_det_log (what the device uses).
 * Returns 0, or -1 on invalid arguments.
 */
int bso_grid_dims(const int32_t extent[3], int32_t bin, int32_t* width, int32_t* height);
int bso_grid_picture(const int32_t* xyz, int64_t n, const int32_t extent[3], int32_t bin, int32_t bin_height,
                           int libm_log, double* image, double* ground_th);

#ifdef __cplusplus
}
#endif
#endif
