// instruction issue-cost microbench (wave64, 8 independent chains per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed, int iters) {
  uint64_t a[8];
  uint32_t lo[8], hi[8];
  for (int i = 0; i < 8; i++) { a[i] = seed * (i + 3) + threadIdx.x; lo[i] = (uint32_t)a[i]; hi[i] = (uint32_t)(a[i] >> 32) | 1; }
  uint64_t m = seed | 1;
  uint32_t m32 = (uint32_t)seed | 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(m));
      if (MODE == 1) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a[i]) : "v"(lo[i]), "v"(m32) : "s20", "s21");
      if (MODE == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(m32), "v"(m32) : "vcc");
      if (MODE == 5) asm volatile("v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(m), "+v"(lo[i]) : "v"(m32) : "vcc");
      if (MODE == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo[i]) : "v"(hi[i]));
      if (MODE == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 10) asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo[i]) : "v"(m32) : "vcc");
      if (MODE == 11) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(m32), "v"(hi[i]));
      if (MODE == 12) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(m32));
      if (MODE == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(m32), "v"(hi[i]));
      if (MODE == 14) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(m32), "v"(hi[i]));
      if (MODE == 15) asm volatile("v_lshlrev_b64 %0, 13, %0" : "+v"(a[i]));
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i] ^ lo[i] ^ ((uint64_t)hi[i] << 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int M> void run(const char *name, uint64_t *out, int ninstr) {
  const int blocks = 2048, iters = 1024;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters * 8;
  printf("%-28s %7.3f ms  %6.2f lane-cycles per group (%d instr)  [@2.4GHz]\n", name, ms, 256.0 * 128 * 2.4e9 / (ops / (ms * 1e-3)), ninstr);
}
int main() {
  uint64_t *out; hipMalloc(&out, 8 * 2048 * 256);
  run<6>("v_xor_b32", out, 1); run<7>("v_alignbit_b32", out, 1); run<11>("v_bfi_b32", out, 1); run<13>("v_add3_u32", out, 1); run<14>("v_xad_u32", out, 1);
  run<0>("v_lshl_add_u64", out, 1); run<15>("v_lshlrev_b64", out, 1); run<4>("v_add_co+v_addc_co", out, 2); run<5>("v_cmp_lt_u64+cndmask", out, 2); run<10>("v_cmp_eq_u32+cndmask", out, 2);
  run<1>("v_mad_u64_u32", out, 1); run<2>("v_mul_lo_u32", out, 1); run<3>("v_mul_hi_u32", out, 1); run<8>("v_mul_u32_u24", out, 1); run<9>("v_mad_u32_u24", out, 1); run<12>("v_mul_hi_u32_u24", out, 1);
  return 0;
}
