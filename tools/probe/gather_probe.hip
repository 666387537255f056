// Developer micro-benchmark: cycles per dependent "gather step" of one wave, per-lane record loads (9 scattered VMEM
// instructions, the step engine's present shape) against cooperative whole-record loads into LDS (8 lanes per 128-byte
// record, 8 instructions per 64 records) followed by per-lane LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void per_lane(const int4* __restrict__ rec, const int* __restrict__ fl, int mask, int steps, long long* out, int sc1all)
{
  const int lane = threadIdx.x;
  int id = (blockIdx.x * 977 + lane * 131) & mask;
  long long t0 = clock64();
  int acc = 0;
  for (int s = 0; s < steps; s++) {
    const int4* r = rec + (size_t)id * 8;
    v4i a0, a1, a2, a3, a4, a5, a6;
    int f0, f1;
    asm volatile(
        "global_load_dword %7, %9, off sc1\n\t"
        "global_load_dword %8, %10, off sc1\n\t"
        "global_load_dwordx4 %0, %11, off\n\t"
        "global_load_dwordx4 %1, %11, off offset:16\n\t"
        "global_load_dwordx4 %2, %11, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %11, off offset:64\n\t"
        "global_load_dwordx4 %4, %11, off offset:80\n\t"
        "global_load_dwordx4 %5, %11, off offset:96\n\t"
        "global_load_dwordx4 %6, %11, off offset:112\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(f0), "=&v"(f1)
        : "v"(fl + id), "v"(fl + (id ^ 1)), "v"(r)
        : "memory");
    acc += a0.x + a1.y + a2.z + a3.w + f0 + f1;
    id = (a0.x + a3.y + lane * 131 + s) & mask;  // next ids depend on the loaded data
  }
  long long t1 = clock64();
  if (lane == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = acc;
  }
}

__global__ __launch_bounds__(64) void coop_lds(const int4* __restrict__ rec, const int* __restrict__ fl, int mask, int steps, long long* out, int sc1all)
{
  __shared__ __attribute__((aligned(16))) int ids[64];
  __shared__ __attribute__((aligned(16))) int stage[64 * 32];
  const int lane = threadIdx.x;
  int id = (blockIdx.x * 977 + lane * 131) & mask;
  long long t0 = clock64();
  int acc = 0;
  const int grp = lane >> 3, ch = lane & 7;
  for (int s = 0; s < steps; s++) {
    ids[lane] = id;
    // (same wave: LDS operations complete in order)
    const int4 i0 = *reinterpret_cast<const int4*>(ids + grp * 8);
    const int4 i1 = *reinterpret_cast<const int4*>(ids + grp * 8 + 4);
    const int my[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    int f0, f1;
    asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1" : "=&v"(f0), "=&v"(f1) : "v"(fl + id), "v"(fl + (id ^ 1)) : "memory");
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // record of candidate grp*8+i; this lane fetches chunk (ch ^ i) so that the LDS image is swizzled
      const int4* src = rec + (size_t)my[i] * 8 + (ch ^ i);
      // LDS destination: M0 base + lane*16 -> record (grp*8+i)'s image at stage + i*256 ints + grp*32 ints
      if (sc1all)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 256), 16, 0, 16 /*sc1*/);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // candidate `lane` = grp*8 + i with i = lane&7, grp = lane>>3: image at stage + i*256 + grp*32, chunk j at slot j^i
    const int* img = stage + ch * 256 + grp * 32;
    const int4 a0 = *reinterpret_cast<const int4*>(img + ((0 ^ ch) << 2));
    const int4 a1 = *reinterpret_cast<const int4*>(img + ((1 ^ ch) << 2));
    const int4 a2 = *reinterpret_cast<const int4*>(img + ((2 ^ ch) << 2));
    const int4 a3 = *reinterpret_cast<const int4*>(img + ((4 ^ ch) << 2));
    acc += a0.x + a1.y + a2.z + a3.w + f0 + f1;
    id = (a0.x + a3.y + lane * 131 + s) & mask;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = clock64();
  if (lane == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = acc;
  }
}


template <int NL>
__global__ __launch_bounds__(64) void per_lane_n(const int4* __restrict__ rec, int mask, int steps, long long* out)
{
  const int lane = threadIdx.x;
  int id = (blockIdx.x * 977 + lane * 131) & mask;
  long long t0 = clock64();
  int acc = 0;
  for (int s = 0; s < steps; s++) {
    const int4* r = rec + (size_t)id * 8;
    v4i a[NL];
#pragma unroll
    for (int i = 0; i < NL; i++)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[i]) : "v"(r + i) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int sum = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(sum) : "v"(a[i].x));
    acc += sum;
    id = (sum + lane * 131 + s) & mask;
  }
  long long t1 = clock64();
  if (lane == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = acc;
  }
}

int main(int argc, char** argv)
{
  const int steps = 2000;
  for (int logn : {14, 17, 21}) {  // 2 MB (L2 of one XCD), 16 MB, 256 MB (beyond the Infinity Cache's share)
    const size_t n = (size_t)1 << logn;
    int4* rec;
    int* fl;
    long long* out;
    CK(hipMalloc(&rec, n * 128));
    CK(hipMalloc(&fl, n * 4));
    CK(hipMalloc(&out, 65536 * 16));
    std::vector<int> h(n * 32);
    unsigned x = 12345;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      v = (int)(x >> 8);
    }
    CK(hipMemcpy(rec, h.data(), n * 128, hipMemcpyHostToDevice));
    CK(hipMemset(fl, 0, n * 4));
    for (int nl = 1; nl <= 7; nl += 2) {
      for (int blocks : {64, 1024}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0));
          if (nl == 1) per_lane_n<1><<<blocks, 64>>>(rec, (int)n - 1, steps, out);
          if (nl == 3) per_lane_n<3><<<blocks, 64>>>(rec, (int)n - 1, steps, out);
          if (nl == 5) per_lane_n<5><<<blocks, 64>>>(rec, (int)n - 1, steps, out);
          if (nl == 7) per_lane_n<7><<<blocks, 64>>>(rec, (int)n - 1, steps, out);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("records 2^%d blocks %5d per-lane %d x dwordx4: %7.3f us/step = %6.0f cycles at 2.4 GHz\n", logn, blocks, nl, ms * 1e3 / steps, ms * 1e3 / steps * 2400);
      }
    }
    for (int blocks : {64, 256, 2048, 8192}) {
      for (int variant = 0; variant < 3; variant++) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0));
          if (variant == 0)
            per_lane<<<blocks, 64>>>(rec, fl, (int)n - 1, steps, out, 0);
          else
            coop_lds<<<blocks, 64>>>(rec, fl, (int)n - 1, steps, out, variant == 2);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> ho(blocks * 2);
        CK(hipMemcpy(ho.data(), out, blocks * 16, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (int b = 0; b < blocks; b++)
          cyc += (double)ho[b * 2];
        printf("records 2^%d blocks %5d %-22s %8.1f clock64-ticks/step  %7.3f us/step (launch %.2f ms) acc %lld\n", logn, blocks,
               variant == 0 ? "per-lane (9 VMEM)" : variant == 1 ? "cooperative->LDS" : "cooperative->LDS sc1", cyc / blocks / steps,
               ms * 1e3 / steps, ms, ho[1]);
      }
    }
    CK(hipFree(rec));
    CK(hipFree(fl));
    CK(hipFree(out));
  }
  return 0;
}
