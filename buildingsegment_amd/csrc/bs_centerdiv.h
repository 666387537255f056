// Plane-centre division of the region grower, exact and cheap.
//
// The reference divides a wrapping int32 running sum by the plane's size_t point
// count (tmc3/my_function.cpp:249-250, quirk Q3 of DESIGN.md): the int is
// converted to unsigned 64-bit (sign-extended), divided, and the quotient is
// truncated back to int32:
//     c = (int32_t)((uint64_t)(int64_t)c / n)            1 <= n < 2^31
// A generic 64-bit division costs ~110 instructions on CDNA and the grower needs
// three per expansion on its critical path.  Only the LOW 32 bits of the quotient
// survive the truncation, and with N = hi*2^32 + lo (hi = 0 or 0xFFFFFFFF):
//     floor(N / n) mod 2^32 = floor(((hi mod n)*2^32 + lo) / n)
// whose numerator is below n*2^32, i.e. its quotient fits 32 bits.  That quotient
// is estimated in f64 (absolute error far below 1: the reciprocal is good to
// 2^-45 and the quotient is below 2^32) and made exact by ONE integer remainder
// check.  rneg = 0xFFFFFFFF mod n and the reciprocal are shared by the three
// coordinates.  v_rcp_f64 alone is only a ~2^-23 seed (measured: planes with
// negative sums came out one off); one Newton step squares its error.
#pragma once
#include <cstdint>

namespace bs {

struct CenterDiv {
  uint32_t n;
  uint32_t rneg;  // 0xFFFFFFFF mod n
  double rcp;     // 1 / n to ~2^-45 relative
};

#if defined(__HIPCC__)
#define BS_CD_FN __host__ __device__ inline
#else
#define BS_CD_FN inline  // plain C++ (tests/cpp/centerdiv_check.cpp)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define BS_CD_RCP(x) __builtin_amdgcn_rcp(x)
#else
// host stand-in for the hardware seed: float precision, so that the host test
// exercises the same refinement and correction as the device
#define BS_CD_RCP(x) ((double)(1.0f / (float)(x)))
#endif

// floor(num / n) for num < n * 2^32, given the refined reciprocal
BS_CD_FN uint32_t center_div_q(uint32_t hi, uint32_t lo, uint32_t n, double rcp)
{
  const double est = __builtin_fma((double)hi, 4294967296.0, (double)lo) * rcp;  // >= 0, < 2^32 + eps
  uint32_t q = est >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)est;
  const uint64_t num = ((uint64_t)hi << 32) | lo;
  const int64_t rem = (int64_t)(num - (uint64_t)q * n);  // exact; -n <= rem < 2n
  q -= rem < 0 ? 1u : 0u;
  q += rem >= (int64_t)n ? 1u : 0u;
  return q;
}

BS_CD_FN CenterDiv center_div_prepare(uint32_t n)
{
  CenterDiv d;
  d.n = n;
  const double x = (double)n;
  const double r0 = BS_CD_RCP(x);
  d.rcp = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
  d.rneg = 0xFFFFFFFFu - center_div_q(0u, 0xFFFFFFFFu, n, d.rcp) * n;
  return d;
}

// (int32_t)((uint64_t)(int64_t)c / n)
BS_CD_FN int32_t center_div(int32_t c, const CenterDiv& d)
{
  return (int32_t)center_div_q(c < 0 ? d.rneg : 0u, (uint32_t)c, d.n, d.rcp);
}

}  // namespace bs
