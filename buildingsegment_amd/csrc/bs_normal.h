// bs_normal.h -- per-point normal from the hybrid neighbourhood's exact integer
// moment sums (device + host).  Product code (HIP, gfx950).
//
// Restates, in this order, what the reference obtains from Open3D at
// /root/reference/tmc3/my_function.h:63-64:
//   utility::ComputeCovariance  (9 cumulants / n, C = E[ab] - E[a]E[b])
//   FastEigen3x3                (Eberly's robust symmetric 3x3 eigensolver,
//                                smallest eigenvector)
//   EstimateNormals zero-norm fallback -> (0,0,1)
//   OrientNormalsToAlignWithDirection((0,0,1))
// SURVEY.md Appendix A.2/A.3 is the operation-order specification.  The
// neighbourhood sums are accumulated as exact integers on the device; they
// equal the reference's sequential f64 accumulation whenever every partial
// sum is below 2^53 (guaranteed by the |coord| < 2^23 domain check at the ABI).
// Compile with -ffp-contract=off: no expression below may be fused.
#pragma once

#include <stdint.h>

#include "../../include/bs_detmath.h"

namespace bs {

struct V3 {
  double x, y, z;
};

struct Sym3 {
  double a00, a01, a02, a11, a12, a22;
};

BS_HD double dot_tree(const V3& a, const V3& b)
{
  // Eigen's fixed-size reduction order: x0 + (x1 + x2)
  return a.x * b.x + (a.y * b.y + a.z * b.z);
}

BS_HD V3 cross(const V3& a, const V3& b)
{
  V3 o;
  o.x = a.y * b.z - a.z * b.y;
  o.y = a.z * b.x - a.x * b.z;
  o.z = a.x * b.y - a.y * b.x;
  return o;
}

BS_HD double absd(double v) { return __builtin_fabs(v); }

BS_HD V3 eigenvector0(const Sym3& A, double lam)
{
  V3 r0 = {A.a00 - lam, A.a01, A.a02};
  V3 r1 = {A.a01, A.a11 - lam, A.a12};
  V3 r2 = {A.a02, A.a12, A.a22 - lam};
  V3 c01 = cross(r0, r1), c02 = cross(r0, r2), c12 = cross(r1, r2);
  double d0 = dot_tree(c01, c01), d1 = dot_tree(c02, c02), d2 = dot_tree(c12, c12);
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) {
    dmax = d1;
    imax = 1;
  }
  if (d2 > dmax)
    imax = 2;
  V3 best = imax == 0 ? c01 : (imax == 1 ? c02 : c12);
  double len = bs_det_sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
  V3 o = {best.x / len, best.y / len, best.z / len};
  return o;
}

BS_HD V3 eigenvector1(const Sym3& A, const V3& e, double lam)
{
  V3 U, V;
  if (absd(e.x) > absd(e.y)) {
    double inv = 1 / bs_det_sqrt(e.x * e.x + e.z * e.z);
    U.x = -e.z * inv;
    U.y = 0;
    U.z = e.x * inv;
  } else {
    double inv = 1 / bs_det_sqrt(e.y * e.y + e.z * e.z);
    U.x = 0;
    U.y = e.z * inv;
    U.z = -e.y * inv;
  }
  V = cross(e, U);
  V3 AU = {A.a00 * U.x + A.a01 * U.y + A.a02 * U.z, A.a01 * U.x + A.a11 * U.y + A.a12 * U.z,
           A.a02 * U.x + A.a12 * U.y + A.a22 * U.z};
  V3 AV = {A.a00 * V.x + A.a01 * V.y + A.a02 * V.z, A.a01 * V.x + A.a11 * V.y + A.a12 * V.z,
           A.a02 * V.x + A.a12 * V.y + A.a22 * V.z};
  double m00 = U.x * AU.x + U.y * AU.y + U.z * AU.z - lam;
  double m01 = U.x * AV.x + U.y * AV.y + U.z * AV.z;
  double m11 = V.x * AV.x + V.y * AV.y + V.z * AV.z - lam;
  double b00 = absd(m00), b01 = absd(m01), b11 = absd(m11);
  V3 o;
  if (b00 >= b11) {
    double mx = b00 > b01 ? b00 : b01;
    if (mx > 0) {
      if (b00 >= b01) {
        m01 /= m00;
        m00 = 1 / bs_det_sqrt(1 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1 / bs_det_sqrt(1 + m00 * m00);
        m00 *= m01;
      }
      o.x = m01 * U.x - m00 * V.x;
      o.y = m01 * U.y - m00 * V.y;
      o.z = m01 * U.z - m00 * V.z;
      return o;
    }
    return U;
  }
  double mx = b11 > b01 ? b11 : b01;
  if (mx > 0) {
    if (b11 >= b01) {
      m01 /= m11;
      m11 = 1 / bs_det_sqrt(1 + m01 * m01);
      m01 *= m11;
    } else {
      m11 /= m01;
      m01 = 1 / bs_det_sqrt(1 + m11 * m11);
      m11 *= m01;
    }
    o.x = m11 * U.x - m01 * V.x;
    o.y = m11 * U.y - m01 * V.y;
    o.z = m11 * U.z - m01 * V.z;
    return o;
  }
  return U;
}

BS_HD V3 smallest_eigenvector(const Sym3& C)
{
  V3 zero = {0, 0, 0};
  double mc = C.a00;
  mc = C.a01 > mc ? C.a01 : mc;
  mc = C.a02 > mc ? C.a02 : mc;
  mc = C.a11 > mc ? C.a11 : mc;
  mc = C.a12 > mc ? C.a12 : mc;
  mc = C.a22 > mc ? C.a22 : mc;
  if (mc == 0)
    return zero;
  Sym3 A = {C.a00 / mc, C.a01 / mc, C.a02 / mc, C.a11 / mc, C.a12 / mc, C.a22 / mc};
  double off = A.a01 * A.a01 + A.a02 * A.a02 + A.a12 * A.a12;
  if (off > 0) {
    double q = (A.a00 + A.a11 + A.a22) / 3;
    double b00 = A.a00 - q, b11 = A.a11 - q, b22 = A.a22 - q;
    double p = bs_det_sqrt((b00 * b00 + b11 * b11 + b22 * b22 + off * 2) / 6);
    double c00 = b11 * b22 - A.a12 * A.a12;
    double c01 = A.a01 * b22 - A.a12 * A.a02;
    double c02 = A.a01 * A.a12 - b11 * A.a02;
    double det = (b00 * c00 - A.a01 * c01 + A.a02 * c02) / (p * p * p);
    double h = det * 0.5;
    h = h > -1.0 ? h : -1.0;
    h = h < 1.0 ? h : 1.0;
    double angle = bs_det_acos(h) / (double)3;
    const double two_thirds_pi = 2.09439510239319549;
    double beta2 = bs_det_cos(angle) * 2;
    double beta0 = bs_det_cos(angle + two_thirds_pi) * 2;
    double beta1 = -(beta0 + beta2);
    double l0 = q + p * beta0, l1 = q + p * beta1, l2 = q + p * beta2;
    if (h >= 0) {
      V3 v2 = eigenvector0(A, l2);
      if (l2 < l0 && l2 < l1)
        return v2;
      V3 v1 = eigenvector1(A, v2, l1);
      if (l1 < l0 && l1 < l2)
        return v1;
      return cross(v1, v2);
    }
    V3 v0 = eigenvector0(A, l0);
    if (l0 < l1 && l0 < l2)
      return v0;
    V3 v1 = eigenvector1(A, v0, l1);
    if (l1 < l0 && l1 < l2)
      return v1;
    return cross(v0, v1);
  }
  double d0 = A.a00 * mc, d1 = A.a11 * mc, d2 = A.a22 * mc;
  V3 o = {0, 0, 0};
  if (d0 < d1 && d0 < d2)
    o.x = 1;
  else if (d1 < d0 && d1 < d2)
    o.y = 1;
  else
    o.z = 1;
  return o;
}

// Moment sums of the hybrid neighbourhood, exact integers.
struct Moments {
  int64_t sx, sy, sz;
  uint64_t sxx, syy, szz;
  int64_t sxy, sxz, syz;
  int32_t n;
};

BS_HD V3 normal_from_moments(const Moments& m)
{
  Sym3 C;
  if (m.n < 3) {
    C.a00 = 1; C.a11 = 1; C.a22 = 1;
    C.a01 = 0; C.a02 = 0; C.a12 = 0;
  } else {
    double dn = (double)m.n;
    double c0 = (double)m.sx / dn, c1 = (double)m.sy / dn, c2 = (double)m.sz / dn;
    double c3 = (double)m.sxx / dn, c4 = (double)m.sxy / dn, c5 = (double)m.sxz / dn;
    double c6 = (double)m.syy / dn, c7 = (double)m.syz / dn, c8 = (double)m.szz / dn;
    C.a00 = c3 - c0 * c0;
    C.a11 = c6 - c1 * c1;
    C.a22 = c8 - c2 * c2;
    C.a01 = c4 - c0 * c1;
    C.a02 = c5 - c0 * c2;
    C.a12 = c7 - c1 * c2;
  }
  V3 nv = smallest_eigenvector(C);
  if (bs_det_sqrt(dot_tree(nv, nv)) == 0.0) {
    nv.x = 0; nv.y = 0; nv.z = 1;
  }
  if (nv.x * 0.0 + (nv.y * 0.0 + nv.z * 1.0) < 0.0) {
    nv.x *= -1.0;
    nv.y *= -1.0;
    nv.z *= -1.0;
  }
  return nv;
}

}  // namespace bs
