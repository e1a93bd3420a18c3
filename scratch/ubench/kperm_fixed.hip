// kperm_fixed.hip -- Keccak-f throughput of the explicit-register permutation (csrc/gen_keccak_fixed.py): the state lives in
// v[P2_KF_BASE ..] for the whole kernel, the compiler is kept below them with amdgpu_num_vgpr.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../acvm-backend-plonky2_amd/csrc/keccak_fixed.inc"
#define STR2(x) #x
#define STR(x) STR2(x)
// lane i of the state <-> v[BASE + 2i], v[BASE + 2i + 1]
#define KF_SET(i, lo, hi) asm volatile("v_mov_b32 v%c2, %0\n v_mov_b32 v%c3, %1" :: "v"(lo), "v"(hi), "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
#define KF_XOR(i, lo, hi) asm volatile("v_xor_b32 v%c2, v%c2, %0\n v_xor_b32 v%c3, v%c3, %1" :: "v"(lo), "v"(hi), "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
#define KF_GET(i, lo, hi) asm volatile("v_mov_b32 %0, v%c2\n v_mov_b32 %1, v%c3" : "=v"(lo), "=v"(hi) : "n"(P2_KF_BASE + 2 * (i)), "n"(P2_KF_BASE + 2 * (i) + 1))
template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(P2_KF_BASE))) void kperm(uint64_t *out, int iters, uint64_t seed) {
  sfor<0, 25>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const uint64_t v = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
    KF_SET(i, (uint32_t)v, (uint32_t)(v >> 32));
  });
#ifdef STAGGER
  {  // desynchronise the waves of a SIMD: they all run the same B,B,A stream
    const int w = (threadIdx.x >> 6) + 4 * (blockIdx.x & 3);
    for (int i = 0; i < (w * STAGGER) % 16; i++) __builtin_amdgcn_s_sleep(1);
  }
#endif
  for (int it = 0; it < iters; it++) {
    P2_KECCAK_FIXED_PERMUTE();
    KF_XOR(3, (uint32_t)it, 0u);
  }
  uint64_t s = 0;
  sfor<0, 25>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    uint32_t lo, hi;
    KF_GET(i, lo, hi);
    s ^= ((uint64_t)hi << 32) | lo;
  });
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  uint64_t *out; hipMalloc(&out, 8ull * 256 * 8 * 256 * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int gens = argc > 2 ? atoi(argv[2]) : 1;  // generations of blocks: > 1 = block turnover as in the prover's kernels
  for (int K : {1, 2, 3, 4}) {
    const int blocks = 256 * K * gens;
    hipLaunchKernelGGL(kperm, dim3(blocks), dim3(256), 0, 0, out, 2, 12345ull);
    hipDeviceSynchronize();
    const int reps = argc > 4 ? atoi(argv[4]) : 1;  // back-to-back launches inside the timed region
    if (argc > 5) hipLaunchKernelGGL(kperm, dim3(256 * K), dim3(256), 0, 0, out, atoi(argv[5]), 999ull);  // untimed busy launch right before
    hipEventRecord(a);
    for (int r = 1; r < reps; r++) hipLaunchKernelGGL(kperm, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ull);
    const int lds = argc > 3 ? atoi(argv[3]) : 0;   // dynamic LDS bytes per block: caps the blocks per CU (160 KB / lds)
    hipLaunchKernelGGL(kperm, dim3(blocks), dim3(256), lds, 0, out, iters, 12345ull);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("K=%d  %8.3f ms  %7.2f Gperm/s  check %016llx\n", K, ms, (double)iters * blocks * 256 * reps / (ms * 1e-3) / 1e9, (unsigned long long)h[1]);
  }
  return 0;
}
