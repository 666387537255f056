/*
 * bs_oracle.c -- CPU restatement of the buildingSegment hot path (see
 * bs_oracle.h for scope and pinning status).  TEST INFRASTRUCTURE ONLY.
 *
 * Compile: gcc -O2 -fPIC -shared -ffp-contract=off (no -ffast-math).
 * Every floating-point expression below is written in the evaluation order
 * of the reference / of Open3D 0.19 as recalled in SURVEY.md Appendix A.
 */
#include "bs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bs_detmath.h"

double bso_det_acos(double x) { return bs_det_acos(x); }
double bso_det_cos(double x) { return bs_det_cos(x); }

double bso_det_log(double x)
{
  return bs_det_log(x);
}

/* ------------------------------------------------------------------------ */
/* Stage 2 arithmetic: covariance by cumulants + FastEigen3x3 + orientation  */
/* ------------------------------------------------------------------------ */

/* Eigen's fixed-size 3-vector dot/squaredNorm reduce as x0 + (x1 + x2)
 * (redux_novec_unroller splits [0,3) into [0,1) and [1,3)). */
static double dot3_eigen(const double a[3], const double b[3])
{
  return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
}

static void cross3(const double a[3], const double b[3], double o[3])
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Open3D ComputeEigenvector0 (SURVEY Appendix A.3 "eigvec0"). A = {a00,a01,a02,a11,a12,a22} */
static void eigvec0(const double A[6], double l, double out[3])
{
  double r0[3] = {A[0] - l, A[1], A[2]};
  double r1[3] = {A[1], A[3] - l, A[4]};
  double r2[3] = {A[2], A[4], A[5] - l};
  double x01[3], x02[3], x12[3];
  cross3(r0, r1, x01);
  cross3(r0, r2, x02);
  cross3(r1, r2, x12);
  double d0 = dot3_eigen(x01, x01);
  double d1 = dot3_eigen(x02, x02);
  double d2 = dot3_eigen(x12, x12);
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) {
    dmax = d1;
    imax = 1;
  }
  if (d2 > dmax)
    imax = 2;
  const double* x = imax == 0 ? x01 : (imax == 1 ? x02 : x12);
  double s = sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
  out[0] = x[0] / s;
  out[1] = x[1] / s;
  out[2] = x[2] / s;
}

/* Open3D ComputeEigenvector1 (SURVEY Appendix A.3 "eigvec1") */
static void eigvec1(const double A[6], const double e[3], double l, double out[3])
{
  double U[3], V[3];
  if (fabs(e[0]) > fabs(e[1])) {
    double s = 1 / sqrt(e[0] * e[0] + e[2] * e[2]);
    U[0] = -e[2] * s;
    U[1] = 0;
    U[2] = e[0] * s;
  } else {
    double s = 1 / sqrt(e[1] * e[1] + e[2] * e[2]);
    U[0] = 0;
    U[1] = e[2] * s;
    U[2] = -e[1] * s;
  }
  cross3(e, U, V);
  double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2],
                  A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
  double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2],
                  A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
  double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - l;
  double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
  double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - l;
  double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    double mx = a00 > a01 ? a00 : a01;
    if (mx > 0) {
      if (a00 >= a01) {
        m01 /= m00;
        m00 = 1 / sqrt(1 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1 / sqrt(1 + m00 * m00);
        m00 *= m01;
      }
      for (int i = 0; i < 3; i++)
        out[i] = m01 * U[i] - m00 * V[i];
    } else {
      for (int i = 0; i < 3; i++)
        out[i] = U[i];
    }
  } else {
    double mx = a11 > a01 ? a11 : a01;
    if (mx > 0) {
      if (a11 >= a01) {
        m01 /= m11;
        m11 = 1 / sqrt(1 + m01 * m01);
        m01 *= m11;
      } else {
        m11 /= m01;
        m01 = 1 / sqrt(1 + m11 * m11);
        m11 *= m01;
      }
      for (int i = 0; i < 3; i++)
        out[i] = m11 * U[i] - m01 * V[i];
    } else {
      for (int i = 0; i < 3; i++)
        out[i] = U[i];
    }
  }
}

/* Open3D FastEigen3x3 (fast_normal_computation = true), SURVEY Appendix A.3 */
void bso_fast_eigen3x3(const double c[6], double out[3])
{
  double m = c[0];
  for (int i = 1; i < 6; i++)
    if (c[i] > m)
      m = c[i];
  if (m == 0) {
    out[0] = out[1] = out[2] = 0;
    return;
  }
  double A[6];
  for (int i = 0; i < 6; i++)
    A[i] = c[i] / m;
  double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
  if (norm > 0) {
    double q = (A[0] + A[3] + A[5]) / 3;
    double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
    double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
    double c00 = b11 * b22 - A[4] * A[4];
    double c01 = A[1] * b22 - A[4] * A[2];
    double c02 = A[1] * A[4] - b11 * A[2];
    double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
    double h = det * 0.5;
    h = h > -1.0 ? h : -1.0; /* std::min(std::max(h, -1.0), 1.0) */
    h = h < 1.0 ? h : 1.0;
    double angle = bs_det_acos(h) / (double)3;
    const double two_thirds_pi = 2.09439510239319549;
    double beta2 = bs_det_cos(angle) * 2;
    double beta0 = bs_det_cos(angle + two_thirds_pi) * 2;
    double beta1 = -(beta0 + beta2);
    double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
    double v0[3], v1[3], v2[3];
    if (h >= 0) {
      eigvec0(A, e2, v2);
      if (e2 < e0 && e2 < e1) {
        memcpy(out, v2, sizeof v2);
        return;
      }
      eigvec1(A, v2, e1, v1);
      if (e1 < e0 && e1 < e2) {
        memcpy(out, v1, sizeof v1);
        return;
      }
      cross3(v1, v2, out);
    } else {
      eigvec0(A, e0, v0);
      if (e0 < e1 && e0 < e2) {
        memcpy(out, v0, sizeof v0);
        return;
      }
      eigvec1(A, v0, e1, v1);
      if (e1 < e0 && e1 < e2) {
        memcpy(out, v1, sizeof v1);
        return;
      }
      cross3(v0, v1, out);
    }
  } else {
    /* diagonal: compare A*m diagonals */
    double d0 = A[0] * m, d1 = A[3] * m, d2 = A[5] * m;
    out[0] = out[1] = out[2] = 0;
    if (d0 < d1 && d0 < d2)
      out[0] = 1;
    else if (d1 < d0 && d1 < d2)
      out[1] = 1;
    else
      out[2] = 1;
  }
}

/* ComputeCovariance (9 cumulants, list order) + FastEigen3x3 + zero-norm
 * fallback + OrientNormalsToAlignWithDirection((0,0,1)).
 * Call sites: /root/reference/tmc3/my_function.h:63-64. */
void bso_normal_from_list(const int32_t* xyz, const int32_t* idx, int cnt, double out[3])
{
  double cov[6];
  if (cnt < 3) {
    cov[0] = cov[3] = cov[5] = 1;
    cov[1] = cov[2] = cov[4] = 0;
  } else {
    double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < cnt; j++) {
      const int32_t* p = xyz + 3 * (int64_t)idx[j];
      double x = p[0], y = p[1], z = p[2];
      cu[0] += x;
      cu[1] += y;
      cu[2] += z;
      cu[3] += x * x;
      cu[4] += x * y;
      cu[5] += x * z;
      cu[6] += y * y;
      cu[7] += y * z;
      cu[8] += z * z;
    }
    double dn = (double)cnt;
    for (int j = 0; j < 9; j++)
      cu[j] /= dn;
    cov[0] = cu[3] - cu[0] * cu[0];
    cov[3] = cu[6] - cu[1] * cu[1];
    cov[5] = cu[8] - cu[2] * cu[2];
    cov[1] = cu[4] - cu[0] * cu[1];
    cov[2] = cu[5] - cu[0] * cu[2];
    cov[4] = cu[7] - cu[1] * cu[2];
  }
  double nv[3];
  bso_fast_eigen3x3(cov, nv);
  /* EstimateNormals: if (normal.norm() == 0) normal = (0,0,1) */
  if (sqrt(dot3_eigen(nv, nv)) == 0.0) {
    nv[0] = 0;
    nv[1] = 0;
    nv[2] = 1;
  }
  /* OrientNormalsToAlignWithDirection((0,0,1)): norm==0 -> ref; dot<0 -> *= -1 */
  if (nv[0] * 0.0 + (nv[1] * 0.0 + nv[2] * 1.0) < 0.0) {
    nv[0] *= -1.0;
    nv[1] *= -1.0;
    nv[2] *= -1.0;
  }
  out[0] = nv[0];
  out[1] = nv[1];
  out[2] = nv[2];
}

/* ------------------------------------------------------------------------ */
/* Stage 1: exact kNN on a hashed uniform grid (canonical (d2, idx) order)   */
/* ------------------------------------------------------------------------ */

typedef struct {
  int64_t n;
  int64_t cell;
  int64_t mn[3];
  int64_t dim[3];
  int32_t* order;   /* point ids sorted by cell key, ascending id inside a cell */
  int64_t* cstart;  /* [ncell+1] */
  int64_t ncell;
  uint64_t* hkeys;  /* open addressing; ~0 = empty */
  int64_t* hvals;
  uint64_t hmask;
} grid_t;

static uint64_t cell_key(int64_t cx, int64_t cy, int64_t cz)
{
  return (uint64_t)cx | ((uint64_t)cy << 21) | ((uint64_t)cz << 42);
}

static uint64_t hash64(uint64_t k) { return (k * 0x9E3779B97F4A7C15ull) ^ (k >> 29); }

static void grid_free(grid_t* g)
{
  free(g->order);
  free(g->cstart);
  free(g->hkeys);
  free(g->hvals);
  memset(g, 0, sizeof *g);
}

/* stable LSD radix sort of (key, val) pairs, 63-bit keys */
static int radix_sort_pairs(uint64_t* keys, int32_t* vals, int64_t n)
{
  uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
  int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!k2 || !v2) {
    free(k2);
    free(v2);
    return -1;
  }
  uint64_t ormask = 0;
  for (int64_t i = 0; i < n; i++)
    ormask |= keys[i];
  uint64_t *src = keys, *dst = k2;
  int32_t *vs = vals, *vd = v2;
  for (int shift = 0; shift < 64; shift += 11) {
    if (((ormask >> shift) & 0x7FF) == 0 && (ormask >> shift) == 0)
      break;
    int64_t cnt[2049];
    memset(cnt, 0, sizeof cnt);
    for (int64_t i = 0; i < n; i++)
      cnt[((src[i] >> shift) & 0x7FF) + 1]++;
    for (int b = 0; b < 2048; b++)
      cnt[b + 1] += cnt[b];
    for (int64_t i = 0; i < n; i++) {
      int64_t d = cnt[(src[i] >> shift) & 0x7FF]++;
      dst[d] = src[i];
      vd[d] = vs[i];
    }
    uint64_t* t = src;
    src = dst;
    dst = t;
    int32_t* tv = vs;
    vs = vd;
    vd = tv;
  }
  if (src != keys) {
    memcpy(keys, src, sizeof(uint64_t) * (size_t)n);
    memcpy(vals, vs, sizeof(int32_t) * (size_t)n);
  }
  free(k2);
  free(v2);
  return 0;
}

static int grid_build(grid_t* g, const int32_t* xyz, int64_t n, int64_t cell)
{
  memset(g, 0, sizeof *g);
  g->n = n;
  int64_t mx[3];
  for (int a = 0; a < 3; a++) {
    g->mn[a] = xyz[a];
    mx[a] = xyz[a];
  }
  for (int64_t i = 1; i < n; i++)
    for (int a = 0; a < 3; a++) {
      int64_t v = xyz[3 * i + a];
      if (v < g->mn[a])
        g->mn[a] = v;
      if (v > mx[a])
        mx[a] = v;
    }
  /* keep every axis below 2^21 cells */
  for (int a = 0; a < 3; a++)
    while ((mx[a] - g->mn[a]) / cell + 1 >= (1 << 21))
      cell *= 2;
  g->cell = cell;
  for (int a = 0; a < 3; a++)
    g->dim[a] = (mx[a] - g->mn[a]) / cell + 1;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  g->order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!keys || !g->order) {
    free(keys);
    return -1;
  }
  for (int64_t i = 0; i < n; i++) {
    keys[i] = cell_key((xyz[3 * i] - g->mn[0]) / cell, (xyz[3 * i + 1] - g->mn[1]) / cell,
                       (xyz[3 * i + 2] - g->mn[2]) / cell);
    g->order[i] = (int32_t)i;
  }
  if (radix_sort_pairs(keys, g->order, n) != 0) {
    free(keys);
    return -1;
  }
  int64_t nc = 0;
  for (int64_t i = 0; i < n; i++)
    if (i == 0 || keys[i] != keys[i - 1])
      nc++;
  g->ncell = nc;
  g->cstart = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nc + 1));
  uint64_t hs = 16;
  while (hs < (uint64_t)nc * 2)
    hs <<= 1;
  g->hmask = hs - 1;
  g->hkeys = (uint64_t*)malloc(sizeof(uint64_t) * hs);
  g->hvals = (int64_t*)malloc(sizeof(int64_t) * hs);
  if (!g->cstart || !g->hkeys || !g->hvals) {
    free(keys);
    return -1;
  }
  memset(g->hkeys, 0xFF, sizeof(uint64_t) * hs);
  int64_t c = 0;
  for (int64_t i = 0; i < n; i++)
    if (i == 0 || keys[i] != keys[i - 1]) {
      g->cstart[c] = i;
      uint64_t h = hash64(keys[i]) & g->hmask;
      while (g->hkeys[h] != ~0ull)
        h = (h + 1) & g->hmask;
      g->hkeys[h] = keys[i];
      g->hvals[h] = c;
      c++;
    }
  g->cstart[nc] = n;
  free(keys);
  return 0;
}

static int64_t grid_lookup(const grid_t* g, int64_t cx, int64_t cy, int64_t cz)
{
  uint64_t k = cell_key(cx, cy, cz);
  uint64_t h = hash64(k) & g->hmask;
  while (g->hkeys[h] != ~0ull) {
    if (g->hkeys[h] == k)
      return g->hvals[h];
    h = (h + 1) & g->hmask;
  }
  return -1;
}

/* choose a cell edge giving a handful of points per occupied cell */
static int64_t auto_cell(const int32_t* xyz, int64_t n, double radius)
{
  int64_t cell = (int64_t)ceil(radius);
  if (cell < 1)
    cell = 1;
  for (int it = 0; it < 4; it++) {
    grid_t g;
    if (grid_build(&g, xyz, n, cell) != 0)
      return cell;
    double occ = (double)n / (double)g.ncell;
    int64_t used = g.cell;
    grid_free(&g);
    cell = used;
    if (occ >= 3.0 && occ <= 12.0)
      break;
    double f = sqrt(6.0 / occ);
    if (f > 4.0)
      f = 4.0;
    if (f < 0.25)
      f = 0.25;
    int64_t nc = (int64_t)floor((double)cell * f + 0.5);
    if (nc < 1)
      nc = 1;
    if (nc == cell)
      break;
    cell = nc;
  }
  return cell;
}

typedef struct {
  uint64_t d2;
  int32_t idx;
} cand_t;

static inline int cand_less(uint64_t d2a, int32_t ia, uint64_t d2b, int32_t ib)
{
  return d2a < d2b || (d2a == d2b && ia < ib);
}

static inline void topk_insert(cand_t* a, int* cnt, int cap, uint64_t d2, int32_t idx)
{
  int c = *cnt;
  if (c == cap) {
    if (!cand_less(d2, idx, a[cap - 1].d2, a[cap - 1].idx))
      return;
    c = cap - 1;
  }
  int j = c;
  while (j > 0 && cand_less(d2, idx, a[j - 1].d2, a[j - 1].idx)) {
    a[j] = a[j - 1];
    j--;
  }
  a[j].d2 = d2;
  a[j].idx = idx;
  *cnt = c + 1;
}

int bso_knn_normals(const int32_t* xyz, int64_t n, int64_t q0, int64_t q1, int k, double radius,
                    int max_nn, int cell, int32_t* neigh, double* normals)
{
  if (!xyz || n <= 0 || k < 1 || k > 64 || n < k || q0 < 0 || q1 > n || q0 > q1)
    return -1;
  if (normals && (max_nn < 1 || max_nn > 64))
    return -1;
  grid_t g;
  int64_t c = cell > 0 ? cell : auto_cell(xyz, n, radius);
  if (grid_build(&g, xyz, n, c) != 0)
    return -3;
  const double r2 = radius * radius;
  cand_t kb[64], mb[64];
  int32_t midx[64];
  for (int64_t q = q0; q < q1; q++) {
    const int64_t qx = xyz[3 * q], qy = xyz[3 * q + 1], qz = xyz[3 * q + 2];
    const int64_t ci[3] = {(qx - g.mn[0]) / g.cell, (qy - g.mn[1]) / g.cell,
                           (qz - g.mn[2]) / g.cell};
    const int64_t qq[3] = {qx, qy, qz};
    int kc = 0, mc = 0;
    for (int64_t rho = 0;; rho++) {
      if (rho > 8) {
        /* isolated query: rings get expensive, a full scan is just as exact */
        kc = 0;
        mc = 0;
        for (int64_t j = 0; j < n; j++) {
          int64_t ddx = xyz[3 * j] - qx, ddy = xyz[3 * j + 1] - qy, ddz = xyz[3 * j + 2] - qz;
          uint64_t d2 = (uint64_t)(ddx * ddx) + (uint64_t)(ddy * ddy) + (uint64_t)(ddz * ddz);
          topk_insert(kb, &kc, k, d2, (int32_t)j);
          if (normals && (double)d2 < r2)
            topk_insert(mb, &mc, max_nn, d2, (int32_t)j);
        }
        break;
      }
      for (int64_t dz = -rho; dz <= rho; dz++) {
        int64_t cz = ci[2] + dz;
        if (cz < 0 || cz >= g.dim[2])
          continue;
        for (int64_t dy = -rho; dy <= rho; dy++) {
          int64_t cy = ci[1] + dy;
          if (cy < 0 || cy >= g.dim[1])
            continue;
          int shell = (dz == -rho || dz == rho || dy == -rho || dy == rho);
          int64_t step = shell ? 1 : (rho > 0 ? 2 * rho : 1);
          for (int64_t dx = -rho; dx <= rho; dx += step) {
            int64_t cx = ci[0] + dx;
            if (cx < 0 || cx >= g.dim[0])
              continue;
            int64_t cid = grid_lookup(&g, cx, cy, cz);
            if (cid < 0)
              continue;
            for (int64_t s = g.cstart[cid]; s < g.cstart[cid + 1]; s++) {
              int32_t j = g.order[s];
              int64_t ddx = xyz[3 * (int64_t)j] - qx, ddy = xyz[3 * (int64_t)j + 1] - qy,
                      ddz = xyz[3 * (int64_t)j + 2] - qz;
              uint64_t d2 = (uint64_t)(ddx * ddx) + (uint64_t)(ddy * ddy) + (uint64_t)(ddz * ddz);
              topk_insert(kb, &kc, k, d2, j);
              if (normals && (double)d2 < r2)
                topk_insert(mb, &mc, max_nn, d2, j);
            }
          }
        }
      }
      /* guaranteed radius: every point outside the (2rho+1)^3 block is at
       * distance >= R from the query */
      uint64_t R = ~0ull;
      int bounded = 0;
      for (int a = 0; a < 3; a++) {
        if (ci[a] - rho > 0) {
          int64_t lo = g.mn[a] + (ci[a] - rho) * g.cell;
          uint64_t d = (uint64_t)(qq[a] - lo + 1);
          if (d < R)
            R = d;
          bounded = 1;
        }
        if (ci[a] + rho < g.dim[a] - 1) {
          int64_t hi = g.mn[a] + (ci[a] + rho + 1) * g.cell - 1;
          uint64_t d = (uint64_t)(hi + 1 - qq[a]);
          if (d < R)
            R = d;
          bounded = 1;
        }
      }
      if (!bounded)
        break; /* whole cloud examined */
      uint64_t R2 = R * R; /* R < 2^32 */
      int knn_ok = (kc == k) && (kb[k - 1].d2 < R2);
      int nrm_ok = !normals || ((double)R2 >= r2);
      if (knn_ok && nrm_ok)
        break;
    }
    int32_t* row = neigh + (q - q0) * (int64_t)k;
    for (int j = 0; j < k; j++)
      row[j] = kb[j].idx;
    if (normals) {
      for (int j = 0; j < mc; j++)
        midx[j] = mb[j].idx;
      bso_normal_from_list(xyz, midx, mc, normals + 3 * (q - q0));
    }
  }
  grid_free(&g);
  return 0;
}

int bso_knn_brute(const int32_t* xyz, int64_t n, int64_t q0, int64_t q1, int k, int32_t* neigh)
{
  if (!xyz || n <= 0 || k < 1 || k > 64 || n < k)
    return -1;
  cand_t kb[64];
  for (int64_t q = q0; q < q1; q++) {
    int kc = 0;
    for (int64_t j = 0; j < n; j++) {
      int64_t dx = (int64_t)xyz[3 * j] - xyz[3 * q], dy = (int64_t)xyz[3 * j + 1] - xyz[3 * q + 1],
              dz = (int64_t)xyz[3 * j + 2] - xyz[3 * q + 2];
      uint64_t d2 = (uint64_t)(dx * dx) + (uint64_t)(dy * dy) + (uint64_t)(dz * dz);
      topk_insert(kb, &kc, k, d2, (int32_t)j);
    }
    for (int j = 0; j < k; j++)
      neigh[(q - q0) * (int64_t)k + j] = kb[j].idx;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Stage 3: region growing (my_function.cpp:180-258), Appendix B.4 form      */
/* ------------------------------------------------------------------------ */

typedef struct {
  int32_t* v;
  int64_t n, cap;
} ivec;

static int ivec_push(ivec* a, int32_t x)
{
  if (a->n == a->cap) {
    int64_t nc = a->cap ? a->cap * 2 : 1024;
    int32_t* p = (int32_t*)realloc(a->v, sizeof(int32_t) * (size_t)nc);
    if (!p)
      return -1;
    a->v = p;
    a->cap = nc;
  }
  a->v[a->n++] = x;
  return 0;
}

/* owner (nullable) [n]: index of the seed attempt that left the point labelled (-1: unlabelled).  Not a
 * reference output: plane_idx[p] == 1 + #(committed seeds < owner[p]) (cur_planeId only advances on commit,
 * my_function.cpp:199-202), which is how a sharded run turns per-shard results into the global ids. */
static int region_grow_core(const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                            int k, int th_thickness, int th_point_count, double cos_th,
                            int32_t* plane_idx, bso_planes* planes, int64_t* n_seed_attempts, int32_t* owner)
{
  if (!xyz || !normals || !neigh || !plane_idx || n <= 0 || k < 1 || n < k)
    return -1;
  for (int64_t i = 0; i < n; i++)
    plane_idx[i] = -1; /* my_function.h:103 */
  if (owner)
    for (int64_t i = 0; i < n; i++)
      owner[i] = -1;
  ivec list = {0, 0, 0};  /* cur_plane.pointIdx */
  ivec stack = {0, 0, 0}; /* pending Broad(id, depth+1) calls, LIFO */
  ivec all = {0, 0, 0};   /* committed lists, concatenated */
  ivec ids = {0, 0, 0};
  int64_t* offs = NULL;
  double* pn = NULL;
  int32_t* pc = NULL;
  int64_t np = 0, pcap = 0, attempts = 0;
  int cur_plane_id = 1; /* my_function.h:119 */
  int rc = 0;
  int32_t sel[64];

  for (int64_t i = 0; i < n && rc == 0; i++) {
    if (plane_idx[i] != -1) /* my_function.cpp:185 */
      continue;
    attempts++;
    /* :187-191 -- seed goes into pointIdx, its label is NOT set (quirk Q1) */
    double cn[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
    int32_t cc[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    list.n = 0;
    stack.n = 0;
    if (ivec_push(&list, (int32_t)i)) {
      rc = -3;
      break;
    }
    /* running sums over pointIdx in append order; the reference re-sums from
     * {0,0,0} (:241-248), so start at +0.0 + normal[seed] */
    double S[3] = {0.0 + normals[3 * i], 0.0 + normals[3 * i + 1], 0.0 + normals[3 * i + 2]};
    uint32_t C[3] = {(uint32_t)xyz[3 * i], (uint32_t)xyz[3 * i + 1], (uint32_t)xyz[3 * i + 2]};
    int64_t cur = i;
    int depth0 = 1;
    int failed = 0;
    for (;;) {
      /* ---- Broad(cur, depth) :220-258 ---- */
      const int32_t* row = neigh + cur * (int64_t)k;
      int ns = 0;
      for (int t = 1; t < k; t++) { /* slot 0 skipped (Q5) */
        int32_t id = row[t];
        if (plane_idx[id] <= 0) { /* Q4 */
          int32_t px = (int32_t)((uint32_t)xyz[3 * (int64_t)id] - (uint32_t)cc[0]);
          int32_t py = (int32_t)((uint32_t)xyz[3 * (int64_t)id + 1] - (uint32_t)cc[1]);
          int32_t pz = (int32_t)((uint32_t)xyz[3 * (int64_t)id + 2] - (uint32_t)cc[2]);
          double dist = fabs(px * cn[0] + py * cn[1] + pz * cn[2]);
          const double* m = normals + 3 * (int64_t)id;
          if (dist <= (double)th_thickness && cn[0] * m[0] + cn[1] * m[1] + cn[2] * m[2] >= cos_th) {
            sel[ns++] = id;
            if (ivec_push(&list, id)) {
              rc = -3;
              break;
            }
            plane_idx[id] = cur_plane_id;
            if (owner)
              owner[id] = (int32_t)i;
            S[0] += m[0];
            S[1] += m[1];
            S[2] += m[2];
            C[0] += (uint32_t)xyz[3 * (int64_t)id];
            C[1] += (uint32_t)xyz[3 * (int64_t)id + 1];
            C[2] += (uint32_t)xyz[3 * (int64_t)id + 2];
          }
        }
      }
      if (rc)
        break;
      if (depth0 && ns < k - 1) { /* :238-239, quirk Q2: labels stay */
        failed = 1;
        break;
      }
      depth0 = 0;
      /* :241-250 */
      double nrm = sqrt((S[0] * S[0]) + (S[1] * S[1]) + (S[2] * S[2]));
      cn[0] = S[0] / nrm;
      cn[1] = S[1] / nrm;
      cn[2] = S[2] / nrm;
      uint64_t cnt = (uint64_t)list.n;
      for (int a = 0; a < 3; a++) /* int /= size_t : quirk Q3 */
        cc[a] = (int32_t)((uint64_t)(int64_t)(int32_t)C[a] / cnt);
      /* :252-255 recursion in selection order == push reversed, pop */
      for (int t = ns - 1; t >= 0; t--)
        if (ivec_push(&stack, sel[t])) {
          rc = -3;
          break;
        }
      if (rc || stack.n == 0)
        break;
      cur = stack.v[--stack.n];
    }
    if (rc)
      break;
    if (failed)
      continue; /* :193-194 */
    if ((uint64_t)list.n > (uint64_t)(int64_t)th_point_count) { /* :199 */
      if (np == pcap) {
        pcap = pcap ? pcap * 2 : 64;
        offs = (int64_t*)realloc(offs, sizeof(int64_t) * (size_t)(pcap + 1));
        pn = (double*)realloc(pn, sizeof(double) * 3 * (size_t)pcap);
        pc = (int32_t*)realloc(pc, sizeof(int32_t) * 3 * (size_t)pcap);
        if (!offs || !pn || !pc) {
          rc = -3;
          break;
        }
      }
      offs[np] = all.n;
      for (int64_t t = 0; t < list.n; t++)
        if (ivec_push(&all, list.v[t])) {
          rc = -3;
          break;
        }
      if (ivec_push(&ids, cur_plane_id))
        rc = -3;
      for (int a = 0; a < 3; a++) {
        pn[3 * np + a] = cn[a];
        pc[3 * np + a] = cc[a];
      }
      np++;
      cur_plane_id++;
    } else {
      for (int64_t t = 0; t < list.n; t++) { /* :203-208 */
        plane_idx[list.v[t]] = -1;
        if (owner)
          owner[list.v[t]] = -1;
      }
    }
  }
  free(list.v);
  free(stack.v);
  if (n_seed_attempts)
    *n_seed_attempts = attempts;
  if (rc) {
    free(all.v);
    free(ids.v);
    free(offs);
    free(pn);
    free(pc);
    return rc;
  }
  if (planes) {
    if (!offs)
      offs = (int64_t*)malloc(sizeof(int64_t));
    offs[np] = all.n;
    planes->n_planes = (int32_t)np;
    planes->id = ids.v;
    planes->normal = pn;
    planes->center = pc;
    planes->offset = offs;
    planes->point_idx = all.v;
  } else {
    free(all.v);
    free(ids.v);
    free(offs);
    free(pn);
    free(pc);
  }
  return 0;
}

int bso_region_grow(const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                    int k, int th_thickness, int th_point_count, double cos_th,
                    int32_t* plane_idx, bso_planes* planes, int64_t* n_seed_attempts)
{
  return region_grow_core(xyz, normals, neigh, n, k, th_thickness, th_point_count, cos_th, plane_idx, planes,
                          n_seed_attempts, NULL);
}

int bso_region_grow_owner(const int32_t* xyz, const double* normals, const int32_t* neigh, int64_t n,
                          int k, int th_thickness, int th_point_count, double cos_th,
                          int32_t* plane_idx, bso_planes* planes, int64_t* n_seed_attempts, int32_t* owner)
{
  return region_grow_core(xyz, normals, neigh, n, k, th_thickness, th_point_count, cos_th, plane_idx, planes,
                          n_seed_attempts, owner);
}

void bso_planes_free(bso_planes* p)
{
  if (!p)
    return;
  free(p->id);
  free(p->normal);
  free(p->center);
  free(p->offset);
  free(p->point_idx);
  memset(p, 0, sizeof *p);
}

/* ------------------------------------------------------------------------- */
/* 2-D raster (TMC3.cpp:75-76, :123-174, :183-199)                            */
/* ------------------------------------------------------------------------- */
int bso_grid_dims(const int32_t extent[3], int32_t bin, int32_t* width, int32_t* height)
{
  if (!extent || bin <= 0 || extent[0] < 0 || extent[1] < 0)
    return -1;
  *width = extent[0] / bin + 2;  /* TMC3.cpp:75 */
  *height = extent[1] / bin + 2; /* TMC3.cpp:76 */
  return 0;
}

int bso_grid_picture(const int32_t* xyz, int64_t n, const int32_t extent[3], int32_t bin, int32_t bin_height,
                           int libm_log, double* image, double* ground_th)
{
  int32_t width, height;
  if (!xyz || n <= 0 || !image || bin_height <= 0 || bso_grid_dims(extent, bin, &width, &height) != 0 ||
      extent[2] < 0)
    return -1;
  /* groundTH (TMC3.cpp:183-199) */
  const int64_t nb = (int64_t)extent[2] / bin_height + 1;
  int* hist = (int*)calloc((size_t)nb, sizeof(int));
  if (!hist)
    return -1;
  const int TH = (int)(n / 2);
  for (int64_t i = 0; i < n; i++)
    hist[xyz[3 * i + 2] / bin_height]++;
  int total = 0;
  int64_t b;
  for (b = 0; b < nb; b++) {
    total += hist[b];
    if (total > TH)
      break;
  }
  free(hist);
  const double th = (double)(int)(b * bin_height);
  if (ground_th)
    *ground_th = th;
  /* compute_gird_picture (TMC3.cpp:123-174) */
  const int64_t npix = (int64_t)width * height;
  for (int64_t q = 0; q < npix * 3; q++)
    image[q] = 0.0;
  for (int64_t i = 0; i < n; i++) {
    const int32_t px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const int x = px / bin, y = py / bin;
    for (int xi = 0; xi < 2; xi++)
      for (int yi = 0; yi < 2; yi++) {
        if ((double)pz < th)
          continue;
        const double w = 1.0 * px / bin - x;
        const double h = 1.0 * py / bin - y;
        const double s = ((xi == 1) ? w : (1 - w)) * ((yi == 1) ? h : (1 - h));
        double* p = image + ((int64_t)(y + yi) * width + (x + xi)) * 3;
        p[1] += s;
        p[0] += s * pz;
      }
  }
  for (int64_t q = 0; q < npix; q++) {
    double* p = image + q * 3;
    if (p[1] != 0)
      p[0] = p[0] / p[1];
  }
  for (int64_t q = 0; q < npix; q++) {
    double* p = image + q * 3;
    p[1] = libm_log ? log(p[1] + 1) : bs_det_log(p[1] + 1);
    if (p[1] != 0)
      p[1] += 20;
  }
  return 0;
}
