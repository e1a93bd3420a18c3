// p2gpu-cold-floor -- what a fresh process pays to the HIP runtime before ANY library work: runtime start-up
// (hipGetDeviceCount), device context + first stream, the first allocation, the first launch of a trivial kernel (its code
// object is a few hundred bytes), a 3 GB allocation + memset, a 245 MB upload from pageable memory, a page-locked allocation.
// bench.py prints it as `cold_process.hip_floor` beside `p2gpu-prove --timing`, so that the cold-process prove of the library
// (what one invocation of the reference's CLI would pay, prove_action.rs:27-43) can be read against what no HIP program avoids.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
__global__ void k(int *p) { if (p) *p = 1; }
int main() {
  double t0 = now_ms();
  int n = 0;
  (void)hipGetDeviceCount(&n);
  double t1 = now_ms();
  (void)hipSetDevice(0);
  hipStream_t st;
  (void)hipStreamCreate(&st);
  double t2 = now_ms();
  int *d = nullptr;
  (void)hipMalloc((void **)&d, 4);
  double t3 = now_ms();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, d);
  (void)hipStreamSynchronize(st);
  double t4 = now_ms();
  void *big = nullptr;
  (void)hipMalloc(&big, (size_t)3 << 30);
  double t5 = now_ms();
  (void)hipMemsetAsync(big, 0, (size_t)3 << 30, st);
  (void)hipStreamSynchronize(st);
  double t6 = now_ms();
  std::vector<char> host((size_t)245 << 20, 1);
  double t7 = now_ms();
  (void)hipMemcpyAsync(big, host.data(), host.size(), hipMemcpyHostToDevice, st);
  (void)hipStreamSynchronize(st);
  double t8 = now_ms();
  void *pin = nullptr;
  (void)hipHostMalloc(&pin, 64 << 20, hipHostMallocDefault);
  double t9 = now_ms();
  printf("{\"hipGetDeviceCount_ms\": %.2f, \"setdevice_stream_ms\": %.2f, \"first_malloc_ms\": %.2f, \"first_launch_sync_ms\": %.2f, "
         "\"malloc_3GB_ms\": %.2f, \"memset_3GB_ms\": %.2f, \"h2d_245MB_pageable_ms\": %.2f, \"hostmalloc_64MB_ms\": %.2f, \"total_ms\": %.2f}\n",
         t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t8 - t7, t9 - t8, t9 - t0);
  return 0;
}
