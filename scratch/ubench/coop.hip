// latency experiment: one Keccak-f[1600] spread over 25 lanes (one 64-bit lane each) vs one
// lane per permutation, measured as time per dependent chain of permutations
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../acvm-backend-plonky2_amd/csrc/keccak.hpp"
using namespace p2;

__constant__ uint64_t RCc[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl((int)(uint32_t)v, src, 64), hi = __shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rotv(uint64_t x, uint32_t n) {  // variable rotate, n in [0,63]
  return n ? ((x << n) | (x >> (64 - n))) : x;
}
// lane L = x + 5y (L < 25) of a 32-lane half holds A[x][y]; lanes 25..31 idle
__device__ uint64_t coop_keccak(uint64_t a, int L, int base, uint32_t rho_out, int pi_src) {
  const int x = L % 5, y = L / 5;
  for (int r = 0; r < 24; r++) {
    // theta: column parity via 4 shuffles
    uint64_t c = a ^ shfl64(a, base + (L + 5) % 25) ^ shfl64(a, base + (L + 10) % 25) ^ shfl64(a, base + (L + 15) % 25) ^
                 shfl64(a, base + (L + 20) % 25);
    uint64_t cm = shfl64(c, base + (x + 4) % 5 + 5 * y), cp = shfl64(c, base + (x + 1) % 5 + 5 * y);
    a ^= cm ^ rotv(cp, 1);
    // rho + pi: this lane becomes B[x][y] = rot(A[x'][y'], r[x'][y']) with (x', y') = pi^-1(x, y)
    uint64_t src = shfl64(a, base + pi_src);
    uint64_t b = rotv(src, rho_out);
    // chi along the row
    uint64_t b1 = shfl64(b, base + (x + 1) % 5 + 5 * y), b2 = shfl64(b, base + (x + 2) % 5 + 5 * y);
    a = b ^ (~b1 & b2);
    if (L == 0) a ^= RCc[r];
  }
  return a;
}
__global__ void k_coop(uint64_t *io, int chain) {
  const int lane = threadIdx.x & 63, half = lane >> 5, L = lane & 31, base = half * 32;
  // tables: rotation offset of the SOURCE lane feeding this lane, and the source lane index
  const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  int x = L % 5, y = L / 5;
  // B[y'][2x'+3y'] = rot(A[x'][y']): lane (X, Y) receives from (x', y') with y' = X, 2x'+3y' = Y (mod 5) -> x' = (Y - 3X) * 3 mod 5
  int xs = ((y - 3 * x) % 5 + 5) % 5 * 3 % 5, ys = x;
  int pi_src = xs + 5 * ys;
  uint32_t rho_out = L < 25 ? rot[pi_src] : 0;
  if (L >= 25) { pi_src = L; }
  uint64_t a = L < 25 ? io[(blockIdx.x * 2 + half) * 25 + L] : 0;
  for (int i = 0; i < chain; i++) a = coop_keccak(a, L < 25 ? L : 0, base, rho_out, L < 25 ? pi_src : 0);
  if (L < 25) io[(blockIdx.x * 2 + half) * 25 + L] = a;
}
__global__ void k_single(uint64_t *io, int chain) {
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = io[(blockIdx.x * blockDim.x + threadIdx.x) * 25 + i];
  for (int i = 0; i < chain; i++) keccak_f1600(st);
  for (int i = 0; i < 25; i++) io[(blockIdx.x * blockDim.x + threadIdx.x) * 25 + i] = st[i];
}
int main() {
  const int chain = 64;
  uint64_t h[50], h2[50];
  for (int i = 0; i < 50; i++) h[i] = 0x123456789abcdefULL * (i + 1);
  uint64_t *d; hipMalloc(&d, sizeof h);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_single, dim3(1), dim3(2), 0, 0, d, chain); hipDeviceSynchronize();
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipEventRecord(a); hipLaunchKernelGGL(k_single, dim3(1), dim3(2), 0, 0, d, chain); hipEventRecord(b); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b); hipMemcpy(h2, d, sizeof h, hipMemcpyDeviceToHost);
  printf("one lane per permutation : %.2f us per permutation (chain of %d)\n", ms * 1e3 / chain, chain);
  uint64_t ref[50]; memcpy(ref, h2, sizeof ref);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_coop, dim3(1), dim3(64), 0, 0, d, chain); hipDeviceSynchronize();
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipEventRecord(a); hipLaunchKernelGGL(k_coop, dim3(1), dim3(64), 0, 0, d, chain); hipEventRecord(b); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b); hipMemcpy(h2, d, sizeof h, hipMemcpyDeviceToHost);
  printf("25 lanes per permutation : %.2f us per permutation, results %s\n", ms * 1e3 / chain, memcmp(ref, h2, sizeof ref) ? "DIFFER" : "match");
  return 0;
}
