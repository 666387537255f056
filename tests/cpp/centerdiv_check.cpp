// Host-side exhaustive-ish check of buildingsegment_amd/csrc/bs_centerdiv.h against the
// reference expression (int32_t)((uint64_t)(int64_t)c / n) (tmc3/my_function.cpp:249-250).
#include "bs_centerdiv.h"

#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv)
{
  const long iters = argc > 1 ? atol(argv[1]) : 20000000;
  std::mt19937_64 r(12345);
  long bad = 0, tot = 0;
  auto chk = [&](int32_t c, uint32_t n) {
    const bs::CenterDiv d = bs::center_div_prepare(n);
    const int32_t got = bs::center_div(c, d);
    const int32_t ref = (int32_t)((uint64_t)(int64_t)c / (uint64_t)n);
    tot++;
    if (got != ref) {
      if (bad < 10)
        printf("BAD c=%d n=%u got=%d ref=%d\n", c, n, got, ref);
      bad++;
    }
  };
  const int32_t cs[] = {0, 1, -1, 2, -2, 2147483647, -2147483647 - 1, -2147483647, 1000, -1000, 65536, -65536, 65535, -65535};
  const uint32_t ns[] = {1, 2, 3, 4, 5, 7, 255, 256, 257, 65535, 65536, 65537, 1000000, 2147483647u, 2147483646u,
                         1073741824u, 1073741823u, 3000, 108197};
  for (int32_t c : cs)
    for (uint32_t n : ns)
      chk(c, n);
  for (uint32_t n = 1; n < 3000; n++)  // small planes: every n, values around the int32 wrap
    for (int d = -3; d <= 3; d++) {
      chk((int32_t)(0x7FFFFFFFu + (uint32_t)d), n);
      chk((int32_t)d, n);
      chk((int32_t)(n * 1000u + (uint32_t)d), n);
    }
  for (long i = 0; i < iters; i++) {
    int32_t c = (int32_t)r();
    const uint64_t x = r();
    uint32_t n;
    switch (x & 3) {
      case 0: n = 1 + (x >> 8) % 1000; break;
      case 1: n = 1 + (x >> 8) % 200000; break;
      case 2: n = 1 + (x >> 8) % 2147483647u; break;
      default: n = 1u << ((x >> 8) % 31); break;
    }
    if ((i & 7) == 0)
      c = (int32_t)(x >> 40) - 8000000;
    chk(c, n);
  }
  printf("checked=%ld bad=%ld\n", tot, bad);
  return bad != 0;
}
