// carry.hip -- what do the carry-chain building blocks of csrc/gl.hpp cost on gfx950?  8 independent chains per
// lane, K waves per SIMD; time per block (ns per wave-block per SIMD slot) for: VOP2 carry pairs through vcc,
// VOP3 pairs through another SGPR pair, the "x -= eps under a borrow" fix as 2 VALU + s_andn2 and as 3 VOP2, and
// whole modular subtractions built from either.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdint>
enum { PAIR_VCC, PAIR_SGPR, FIX_SALU, FIX_VOP2, SUB_SALU, SUB_VOP2, SUB_PORTABLE, NM };
static const char *NAMES[NM] = {"add_co+addc_co (vcc)", "add_co+addc_co (s[20:21])", "fix: 2 VALU + s_andn2", "fix: 3 VOP2",
                                "sub: 4 VALU + s_andn2", "sub: 5 VOP2", "sub: portable (hipcc)"};
static const int NVALU[NM] = {2, 2, 2, 3, 4, 5, 6};
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed, int iters) {
  uint32_t lo[8], hi[8], t[8];
  for (int i = 0; i < 8; i++) {
    uint64_t a = seed * (i + 3) + threadIdx.x * 0x9E3779B97F4A7C15ULL;
    lo[i] = (uint32_t)a; hi[i] = (uint32_t)(a >> 32); t[i] = 0;
  }
  const uint32_t m0 = (uint32_t)seed | 1, m1 = (uint32_t)(seed >> 7) | 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == PAIR_VCC)
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(m0), "v"(m1) : "vcc");
      if (MODE == PAIR_SGPR)
        asm volatile("v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]" : "+v"(lo[i]), "+v"(hi[i]) : "v"(m0), "v"(m1) : "s20", "s21");
      if (MODE == FIX_SALU)
        asm volatile("v_addc_co_u32 %0, s[20:21], %0, 0, vcc\n s_andn2_b64 vcc, vcc, s[20:21]\n v_subb_co_u32 %1, vcc, %1, 0, vcc"
                     : "+v"(lo[i]), "+v"(hi[i]) : : "vcc", "scc", "s20", "s21");
      if (MODE == FIX_VOP2)
        asm volatile("v_subb_co_u32 %2, vcc, %0, %0, vcc\n v_sub_co_u32 %0, vcc, %0, %2\n v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
                     : "+v"(lo[i]), "+v"(hi[i]), "=&v"(t[i]) : : "vcc");
      if (MODE == SUB_SALU)
        asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %3, vcc\n"
                     "v_addc_co_u32 %0, s[20:21], %0, 0, vcc\n s_andn2_b64 vcc, vcc, s[20:21]\n v_subb_co_u32 %1, vcc, %1, 0, vcc"
                     : "+v"(lo[i]), "+v"(hi[i]) : "v"(m0), "v"(m1) : "vcc", "scc", "s20", "s21");
      if (MODE == SUB_VOP2)
        asm volatile("v_sub_co_u32 %0, vcc, %0, %3\n v_subb_co_u32 %1, vcc, %1, %4, vcc\n"
                     "v_subb_co_u32 %2, vcc, %0, %0, vcc\n v_sub_co_u32 %0, vcc, %0, %2\n v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
                     : "+v"(lo[i]), "+v"(hi[i]), "=&v"(t[i]) : "v"(m0), "v"(m1) : "vcc");
      if (MODE == SUB_PORTABLE) {
        uint64_t a = ((uint64_t)hi[i] << 32) | lo[i], b = ((uint64_t)m1 << 32) | m0;
        a = a >= b ? a - b : a - b + 0xFFFFFFFF00000001ULL;
        asm volatile("" : "+v"(a));
        lo[i] = (uint32_t)a; hi[i] = (uint32_t)(a >> 32);
      }
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= lo[i] ^ ((uint64_t)hi[i] << 32) ^ t[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int M>
static void run(uint64_t *out, int K) {
  const int blocks = 256 * K, iters = 4096;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, 64);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double blocks_done = (double)iters * 8 * 64.0 * blocks * 4;  // lane-level executions of the block
  printf("%-28s K=%d  %8.3f ms  %7.3f T blocks/s  %6.2f T lane-VALU/s\n", NAMES[M], K, ms, blocks_done / (ms * 1e-3) / 1e12,
         blocks_done * NVALU[M] / (ms * 1e-3) / 1e12);
}
int main() {
  uint64_t *out;
  hipMalloc(&out, 8ull * 256 * 8 * 256);
  for (int K : {2, 4, 8}) {
    run<PAIR_VCC>(out, K); run<PAIR_SGPR>(out, K); run<FIX_SALU>(out, K); run<FIX_VOP2>(out, K);
    run<SUB_SALU>(out, K); run<SUB_VOP2>(out, K); run<SUB_PORTABLE>(out, K);
    printf("\n");
  }
  return 0;
}
