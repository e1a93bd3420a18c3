// glrate.hip -- throughput of the field primitives of csrc/gl.hpp on gfx950: modmuls, adds and subs per second
// with K waves per SIMD, carry-chain assembly (default) against hipcc's lowering of the portable code
// (-DP2_GL_ASM=0).  ILP chains per lane: -DCHAINS=n (independent dependency chains a wave can interleave).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdint>
#include "../../acvm-backend-plonky2_amd/csrc/gl.hpp"
using namespace p2;
#ifndef CHAINS
#define CHAINS 4
#endif
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed, int iters) {
  gl_t x[CHAINS], y[CHAINS];
  for (int c = 0; c < CHAINS; c++) {
    x[c] = gl_canon_c(seed * (threadIdx.x + 3 + c) + blockIdx.x);
    y[c] = gl_canon_c(seed * (threadIdx.x + 7 + 2 * c) ^ 0x9E3779B97F4A7C15ULL);
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (MODE == 0) x[c] = gl_mul(x[c], y[c]);
      if (MODE == 1) { const gl_t s = gl_add(x[c], y[c]); y[c] = gl_sub(x[c], y[c]); x[c] = s; }
      if (MODE == 2) { const gl_t t = gl_mul(x[c], y[c]); y[c] = gl_sub(x[c], t); x[c] = gl_add(x[c], t); }
    }
  }
  gl_t acc = 0;
  for (int c = 0; c < CHAINS; c++) acc ^= x[c] ^ y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char *name, int K, double ops_per_iter) {
  const int iters = 2000, blocks = 256 * K;
  uint64_t *out;
  hipMalloc(&out, 8ull * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 256 * iters * CHAINS * ops_per_iter;
  printf("%-28s K=%d chains=%d  %8.3f ms  %7.3f T lane-ops/s\n", name, K, CHAINS, ms, ops / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int K : {1, 2, 4, 8}) {
    run<0>("mul", K, 1);
    run<1>("add+sub (butterfly)", K, 1);
    run<2>("mul + butterfly", K, 1);
  }
  return 0;
}
