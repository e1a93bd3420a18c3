// kperm.hip -- Keccak-f[1600] throughput of the product's permutation (keccak.hpp) alone: no loads, no absorb.
// K waves per SIMD through __launch_bounds__; grid = 256 CUs x K blocks of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../acvm-backend-plonky2_amd/csrc/keccak.hpp"
#ifdef KPERM_ASM
#include "kperm_asm.inc"
#endif
using namespace p2;
template <int K>
__global__ __launch_bounds__(256, K) void kperm(uint64_t *out, int iters, uint64_t seed) {
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
  for (int it = 0; it < iters; it++) {
#ifdef KPERM_ASM
    keccak_f1600_asm(st);
#else
    keccak_f1600(st);
#endif
    st[3] ^= it;  // something absorbed, so that consecutive permutations are not one loop nest to the compiler
  }
  uint64_t s = 0;
  for (int i = 0; i < 25; i++) s ^= st[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
static int g_gens = 1;
template <int K>
static void run(uint64_t *out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * K * g_gens;
  hipLaunchKernelGGL(kperm<K>, dim3(blocks), dim3(256), 0, 0, out, 2, 12345ull);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kperm<K>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ull);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  uint64_t h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  const double perms = (double)iters * blocks * 256;
  printf("K=%d  %8.3f ms  %7.2f Gperm/s  check %016llx\n", K, ms, perms / (ms * 1e-3) / 1e9, (unsigned long long)h[1]);
}
int main(int argc, char **argv) {
  if (argc > 2) g_gens = atoi(argv[2]);
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  uint64_t *out; hipMalloc(&out, 8ull * 256 * 8 * 256 * 64);
  run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters); run<5>(out, iters); run<6>(out, iters); run<7>(out, iters);
  return 0;
}
