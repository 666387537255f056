/*
 * bs_detmath.h -- normative, platform-independent acos()/cos() for the
 * normal-estimation stage of the buildingSegment hot path.
 *
 * Why this exists: the reference calls Open3D's FastEigen3x3
 * (call site /root/reference/tmc3/my_function.h:63), which evaluates
 * std::acos / std::cos from the platform libm.  libm results differ in the
 * last bit between MSVC, glibc and the GPU's ocml, so "the reference normal"
 * is already platform-defined.  To make host and device agree bit-for-bit the
 * build fixes ONE evaluation scheme, written only with IEEE-754 binary64
 * + - * / sqrt (all correctly rounded, no FMA contraction: compile with
 * -ffp-contract=off).  Both the CPU oracle (oracle/) and the HIP product
 * (buildingsegment_amd/csrc/) include this header; tests pin it to libm
 * within 2 ulp (tests/test_detmath.py).
 *
 * Algorithms: classic fdlibm-style argument reduction + minimax kernels
 * (rational R(z) for acos, degree-14/13 polynomials for cos/sin).
 * Domain: bs_det_acos: [-1, 1];  bs_det_cos: [0, pi + 0.1].
 */
#ifndef BS_DETMATH_H
#define BS_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BS_HD __host__ __device__ inline
#else
#define BS_HD static inline
#endif

BS_HD double bs_det_sqrt(double x)
{
  /* correctly rounded on x86-64 (sqrtsd) and on gfx950 (LLVM expands
   * llvm.sqrt.f64 to a correctly rounded sequence) */
  return __builtin_sqrt(x);
}

BS_HD double bs_det_clear_low32(double x)
{
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  u &= 0xFFFFFFFF00000000ull;
  memcpy(&x, &u, sizeof u);
  return x;
}

BS_HD uint32_t bs_det_hi32(double x)
{
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  return (uint32_t)(u >> 32);
}

/* rational kernel shared by the three acos ranges */
BS_HD double bs_det_acos_R(double z)
{
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  return p / q;
}

BS_HD double bs_det_acos(double x)
{
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  const double pi = 3.14159265358979311600e+00;
  double ax = x < 0.0 ? -x : x;
  if (ax >= 1.0) { /* callers clamp to [-1,1] */
    if (x > 0.0)
      return 0.0;
    return pi + 2.0 * pio2_lo;
  }
  if (ax < 0.5) {
    if (ax < 6.938893903907228e-18) /* 2^-57 */
      return pio2_hi + pio2_lo;
    double z = x * x;
    double r = bs_det_acos_R(z);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0) {
    double z = (1.0 + x) * 0.5;
    double s = bs_det_sqrt(z);
    double r = bs_det_acos_R(z);
    double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  {
    double z = (1.0 - x) * 0.5;
    double s = bs_det_sqrt(z);
    double df = bs_det_clear_low32(s);
    double c = (z - df * df) / (s + df);
    double r = bs_det_acos_R(z);
    double w = r * s + c;
    return 2.0 * (df + w);
  }
}

/* cos kernel on [-pi/4, pi/4]; (x, y) = head, tail of the reduced argument */
BS_HD double bs_det_kcos(double x, double y)
{
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  uint32_t ix = bs_det_hi32(x) & 0x7fffffffu;
  if (ix < 0x3FD33333u) /* |x| < 0.3 */
    return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ix > 0x3fe90000u) {
    qx = 0.28125;
  } else {
    uint64_t u = (uint64_t)(ix - 0x00200000u) << 32; /* ~|x|/4 */
    memcpy(&qx, &u, sizeof u);
  }
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

/* sin kernel on [-pi/4, pi/4] */
BS_HD double bs_det_ksin(double x, double y)
{
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

BS_HD double bs_det_cos(double x)
{
  /* x in [0, pi + 0.1]: quadrant n in {0,1,2} */
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double pio2_1t = 6.07710050650619224932e-11; /* pi/2 - pio2_1 */
  int n = (int)(x * invpio2 + 0.5);
  double fn = (double)n;
  double r = x - fn * pio2_1; /* exact: pio2_1 has 33 significant bits, n <= 2 */
  double w = fn * pio2_1t;
  double y0 = r - w;
  double y1 = (r - y0) - w;
  switch (n & 3) {
  case 0: return bs_det_kcos(y0, y1);
  case 1: return -bs_det_ksin(y0, y1);
  case 2: return -bs_det_kcos(y0, y1);
  default: return bs_det_ksin(y0, y1);
  }
}

/*
 * bs_det_log: natural logarithm for the density channel of the 2-D raster
 * (reference call site: std::log in buildingSeg::compute_gird_picture,
 * /root/reference/tmc3/TMC3.cpp:165).  Same reasoning as above: libm's log is
 * platform-defined in its last bit, so one scheme is fixed for oracle and device.
 * Classic fdlibm-style evaluation: x = 2^k * (1 + f) with sqrt(2)/2 < 1+f < sqrt(2),
 * s = f / (2 + f), log(1+f) = f - f*f/2 + s*(f*f/2 + R(s*s)), R a degree-7 minimax
 * polynomial in s*s; log(x) = k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f).
 * Domain: finite normal x >= 2^-1022 (the raster only calls it with x >= 1).
 * Within 1 ulp of glibc's log (tests/test_detmath.py).
 */
BS_HD double bs_det_log(double x)
{
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  int32_t hx = (int32_t)(u >> 32);
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i0 = (hx + 0x95f64) & 0x100000; /* 1+f >= sqrt(2): halve it, k += 1 */
  u = ((uint64_t)(uint32_t)(hx | (i0 ^ 0x3ff00000)) << 32) | (u & 0xFFFFFFFFull);
  memcpy(&x, &u, sizeof u);
  k += i0 >> 20;
  const double f = x - 1.0;
  const double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
    if (f == 0.0)
      return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const int32_t i = (hx - 0x6147a) | (0x6b851 - hx);
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

#endif /* BS_DETMATH_H */
