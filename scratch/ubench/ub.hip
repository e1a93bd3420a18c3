// micro-benchmarks: modmul / add / shift-mul / keccak throughput on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../acvm-backend-plonky2_amd/csrc/gl.hpp"
#include "../../acvm-backend-plonky2_amd/csrc/keccak.hpp"
using namespace p2;

template <int MODE>
__global__ __launch_bounds__(256) void k(gl_t *out, gl_t seed, int iters) {
  gl_t a[8];
  gl_t t = seed + threadIdx.x + blockIdx.x * 977;
  for (int i = 0; i < 8; i++) a[i] = gl_canon(t * (i + 3) + i);
  gl_t m = gl_canon(seed * 31 + 7);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) a[i] = gl_mul(a[i], m);
      if (MODE == 1) a[i] = gl_add(a[i], m);
      if (MODE == 2) a[i] = gl_sub(a[i], m);
      if (MODE == 3) { gl_t u = a[i], v = gl_mul(a[(i + 1) & 7], m); a[i] = gl_add(u, v); a[(i + 1) & 7] = gl_sub(u, v); }
      if (MODE == 4) a[i] = gl_mul_small(a[i], 4);
    }
  }
  gl_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kk(uint64_t *out, uint64_t seed, int iters) {
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = seed + i * 77 + threadIdx.x + blockIdx.x * 13;
  for (int it = 0; it < iters; it++) keccak_f1600(st);
  uint64_t s = 0;
  for (int i = 0; i < 25; i++) s ^= st[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> double timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  gl_t *out; hipMalloc(&out, 8 * 2048 * 256);
  const int blocks = 2048, iters = 512;
  const char *names[] = {"gl_mul", "gl_add", "gl_sub", "butterfly(mul+add+sub)", "gl_mul_small"};
  double ops = (double)blocks * 256 * iters * 8;
  double ms;
  ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); }); printf("%-26s %8.3f ms  %8.2f Gop/s  %6.1f lane-cycles/op\n", names[0], ms, ops / ms / 1e6, 256.0*128*2.4e9 / (ops / (ms*1e-3)));
  ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); }); printf("%-26s %8.3f ms  %8.2f Gop/s  %6.1f lane-cycles/op\n", names[1], ms, ops / ms / 1e6, 256.0*128*2.4e9 / (ops / (ms*1e-3)));
  ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); }); printf("%-26s %8.3f ms  %8.2f Gop/s  %6.1f lane-cycles/op\n", names[2], ms, ops / ms / 1e6, 256.0*128*2.4e9 / (ops / (ms*1e-3)));
  ms = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); }); printf("%-26s %8.3f ms  %8.2f Gop/s  %6.1f lane-cycles/op\n", names[3], ms, ops / ms / 1e6, 256.0*128*2.4e9 / (ops / (ms*1e-3)));
  ms = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 12345ull, iters); }); printf("%-26s %8.3f ms  %8.2f Gop/s  %6.1f lane-cycles/op\n", names[4], ms, ops / ms / 1e6, 256.0*128*2.4e9 / (ops / (ms*1e-3)));
  double perms = (double)blocks * 256 * 64;
  ms = timeit([&] { hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, (uint64_t *)out, 999ull, 64); });
  printf("%-26s %8.3f ms  %8.3f Gperm/s  %6.0f lane-cycles/perm\n", "keccak_f1600", ms, perms / ms / 1e6, 256.0*128*2.4e9 / (perms / (ms*1e-3)));
  return 0;
}
