// exec16.hip -- does a wave64 VALU instruction get cheaper when whole 16-lane quarters of EXEC are off?
// One wave (and one wave per SIMD on every CU), a long chain of v_bitop3 / v_alignbit on 8 independent registers,
// with 64, 32, 16, 8 and 1 active lanes.  If the SIMD skipped inactive quarters the time would drop with the lane count.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void k(uint32_t *out, int active, int iters) {
  uint32_t x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 2654435761u + i;
  if ((int)threadIdx.x < active) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 3) & 7]));
        asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(x[(i + 5) & 7]));
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  uint32_t *out;
  hipMalloc(&out, 4 * 64 * 4096);
  const int iters = 20000;
  for (int blocks : {1, 1024, 4096}) {
    for (int active : {64, 32, 16, 8, 1}) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, active, 100);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, active, iters);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("blocks %5d  active lanes %2d  %8.3f ms  %6.2f ns per instruction of the wave\n", blocks, active, ms, ms * 1e6 / (iters * 16.0));
    }
  }
  return 0;
}
