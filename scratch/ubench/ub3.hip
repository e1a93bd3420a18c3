// modmul variants: compiler vs inline-asm 128-bit product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../acvm-backend-plonky2_amd/csrc/gl.hpp"
using namespace p2;

__device__ __forceinline__ void mul128_asm(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t p0, m, p3, cdummy, carry;
  uint32_t r1, r2, r3;
  asm("v_mad_u64_u32 %0, %3, %5, %6, 0\n\t"
      "v_mad_u64_u32 %1, %3, %5, %8, 0\n\t"
      "v_mad_u64_u32 %2, %3, %7, %8, 0\n\t"
      "v_mad_u64_u32 %1, %4, %7, %6, %1"
      : "=&v"(p0), "=&v"(m), "=&v"(p3), "=&s"(cdummy), "=&s"(carry)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  asm("v_add_co_u32 %0, vcc, %3, %4\n\t"
      "v_addc_co_u32 %1, vcc, %5, %6, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %7, 0, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, 0, %8"
      : "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"((uint32_t)(p0 >> 32)), "v"((uint32_t)m), "v"((uint32_t)p3), "v"((uint32_t)(m >> 32)), "v"((uint32_t)(p3 >> 32)),
        "s"(carry)
      : "vcc");
  lo = ((uint64_t)r1 << 32) | (uint32_t)p0;
  hi = ((uint64_t)r3 << 32) | r2;
}
__device__ __forceinline__ gl_t gl_mul_asm(gl_t a, gl_t b) {
  uint64_t lo, hi;
  mul128_asm(a, b, lo, hi);
  return gl_reduce128(lo, hi);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(gl_t *out, gl_t seed, int iters) {
  gl_t a[8];
  gl_t t = seed + threadIdx.x + blockIdx.x * 977;
  for (int i = 0; i < 8; i++) a[i] = gl_canon(t * (i + 3) + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = gl_mul(a[i], a[(i + 3) & 7]);
      if (MODE == 1) a[i] = gl_mul_asm(a[i], a[(i + 3) & 7]);
    }
  }
  gl_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  gl_t *out; hipMalloc(&out, 8 * 2048 * 256);
  gl_t *h = (gl_t *)malloc(8 * 2048 * 256), *h2 = (gl_t *)malloc(8 * 2048 * 256);
  const int blocks = 2048, iters = 512;
  double ops = (double)blocks * 256 * iters * 8;
  for (int mode = 0; mode < 2; mode++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&] { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(mode ? h2 : h, out, 8 * 2048 * 256, hipMemcpyDeviceToHost);
    printf("mode %d: %.3f ms  %.1f lane-cycles/mul\n", mode, ms, 256.0 * 128 * 2.4e9 / (ops / (ms * 1e-3)));
  }
  printf("results equal: %d\n", memcmp(h, h2, 8 * 2048 * 256) == 0);
  return 0;
}
