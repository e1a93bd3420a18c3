// latency experiment, round 3: ONE Keccak-f[1600] spread over 25 lanes of a 32-lane half (two permutations per wave), three
// ds_bpermute stages per round (theta column parity | theta row neighbours | rho-pi-chi fetch of the three rotated lanes),
// against the fixed-register one-lane-per-permutation form the tree tails use today.  Chain of dependent permutations.
//   hipcc -O3 --offload-arch=gfx950 coop2.hip -o coop2 && ./coop2 [blocks] [waves per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../acvm-backend-plonky2_amd/csrc/keccak.hpp"
using namespace p2;

__device__ __forceinline__ uint32_t bperm(uint32_t addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)v); }
__device__ __forceinline__ uint32_t x3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

struct CoopLane {          // per-lane constants of the 25-lane layout (lane L = x + 5y of a 32-lane half)
  uint32_t col[4];         // byte addresses of the four column mates (y + 1..4)
  uint32_t rm, rp;         // row neighbours x - 1, x + 1
  uint32_t b0, b1, b2;     // pi^-1 of (x, y), (x + 1, y), (x + 2, y): the lanes whose ROTATED words chi needs
  uint32_t s;              // v_alignbit shift of this lane's rho rotation
  uint32_t swap;           // 1: exchange the halves before the alignbits (rotation >= 32, or 0)
  uint32_t m0;             // all ones in lane 0 (iota), 0 elsewhere
};
__device__ __forceinline__ CoopLane coop_lane(uint32_t lane) {
  const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  const uint32_t base = lane & 32u, L0 = lane & 31u, L = L0 < 25 ? L0 : 0;
  const int x = L % 5, y = L / 5;
  CoopLane c;
  for (int i = 0; i < 4; i++) c.col[i] = (base + (L + 5 * (i + 1)) % 25) * 4;
  c.rm = (base + (x + 4) % 5 + 5 * y) * 4;
  c.rp = (base + (x + 1) % 5 + 5 * y) * 4;
  // B[X][Y] = rot(A[x'][y']) with y' = X, 2x' + 3y' = Y (mod 5) -> x' = 3 (Y - 3X) mod 5
  auto src = [&](int X, int Y) { int xs = (((Y - 3 * X) % 5 + 5) % 5) * 3 % 5; return (uint32_t)(xs + 5 * X); };
  c.b0 = (base + src(x, y)) * 4;
  c.b1 = (base + src((x + 1) % 5, y)) * 4;
  c.b2 = (base + src((x + 2) % 5, y)) * 4;
  const int r = rot[L];
  c.swap = (r == 0 || r >= 32) ? 1u : 0u;
  c.s = (32 - (r & 31)) & 31;
  c.m0 = L0 == 0 ? 0xFFFFFFFFu : 0u;
  return c;
}
__constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
__device__ __forceinline__ void coop_keccak_f(uint32_t &lo, uint32_t &hi, const CoopLane &c) {
#pragma unroll
  for (int r = 0; r < 24; r++) {
    // theta 1: column parity
    uint32_t t0 = bperm(c.col[0], lo), t1 = bperm(c.col[1], lo), t2 = bperm(c.col[2], lo), t3 = bperm(c.col[3], lo);
    uint32_t u0 = bperm(c.col[0], hi), u1 = bperm(c.col[1], hi), u2 = bperm(c.col[2], hi), u3 = bperm(c.col[3], hi);
    const uint32_t cl = x3(x3(lo, t0, t1), t2, t3), ch = x3(x3(hi, u0, u1), u2, u3);
    // theta 2: D = C[x - 1] ^ rot1(C[x + 1])
    const uint32_t ml = bperm(c.rm, cl), mh = bperm(c.rm, ch), pl = bperm(c.rp, cl), ph = bperm(c.rp, ch);
    lo = x3(lo, ml, __builtin_amdgcn_alignbit(pl, ph, 31));
    hi = x3(hi, mh, __builtin_amdgcn_alignbit(ph, pl, 31));
    // rho in place
    const uint32_t L_ = c.swap ? hi : lo, H_ = c.swap ? lo : hi;
    const uint32_t rl = __builtin_amdgcn_alignbit(L_, H_, c.s), rh = __builtin_amdgcn_alignbit(H_, L_, c.s);
    // pi + chi: this lane becomes B[x][y] ^ (~B[x+1][y] & B[x+2][y]), the B's fetched from where rho left them
    const uint32_t a0 = bperm(c.b0, rl), a1 = bperm(c.b1, rl), a2 = bperm(c.b2, rl);
    const uint32_t h0 = bperm(c.b0, rh), h1 = bperm(c.b1, rh), h2 = bperm(c.b2, rh);
    lo = __builtin_amdgcn_bitop3_b32(a0, a1, a2, 0xD2);
    hi = __builtin_amdgcn_bitop3_b32(h0, h1, h2, 0xD2);
    // iota (lane 0)
    lo ^= (uint32_t)KECCAK_RC[r] & c.m0;
    hi ^= (uint32_t)(KECCAK_RC[r] >> 32) & c.m0;
  }
}

__global__ void k_coop(uint64_t *io, int chain) {
  const uint32_t lane = threadIdx.x & 63u, L = lane & 31u;
  const size_t slot = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const CoopLane c = coop_lane(lane);
  uint64_t a = L < 25 ? io[slot * 25 + L] : 0;
  uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
  for (int i = 0; i < chain; i++) coop_keccak_f(lo, hi, c);
  if (L < 25) io[slot * 25 + L] = ((uint64_t)hi << 32) | lo;
}
// today's tail: one lane per permutation on the fixed registers, lone-wave code placement (PH = 0)
__global__ P2_KF_KERNEL_ATTR void k_single(uint64_t *io, int chain, int nperm) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = k < (size_t)nperm;
  const uint64_t *p = io + (on ? k : 0) * 25;
#define SETI(i) { const uint64_t x = p[i]; P2_KF_SET(i, (uint32_t)x, (uint32_t)(x >> 32)); }
  SETI(0) SETI(1) SETI(2) SETI(3) SETI(4) SETI(5) SETI(6) SETI(7) SETI(8) SETI(9) SETI(10) SETI(11) SETI(12)
  SETI(13) SETI(14) SETI(15) SETI(16) SETI(17) SETI(18) SETI(19) SETI(20) SETI(21) SETI(22) SETI(23) SETI(24)
  for (int i = 0; i < chain; i++) P2_KECCAK_FIXED_PERMUTE_PH(0);
#define GETI(i) { uint32_t lo, hi; P2_KF_GET(i, lo, hi); if (on) io[k * 25 + i] = ((uint64_t)hi << 32) | lo; }
  GETI(0) GETI(1) GETI(2) GETI(3) GETI(4) GETI(5) GETI(6) GETI(7) GETI(8) GETI(9) GETI(10) GETI(11) GETI(12)
  GETI(13) GETI(14) GETI(15) GETI(16) GETI(17) GETI(18) GETI(19) GETI(20) GETI(21) GETI(22) GETI(23) GETI(24)
}

int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1, waves = argc > 2 ? atoi(argv[2]) : 1;
  const int chain = 64;
  const int nperm = blocks * waves * 2;   // permutations the cooperative launch holds (two per wave)
  const size_t words = (size_t)nperm * 25;
  uint64_t *h = (uint64_t *)malloc(words * 8), *h1 = (uint64_t *)malloc(words * 8), *h2 = (uint64_t *)malloc(words * 8);
  for (size_t i = 0; i < words; i++) h[i] = 0x9E3779B97F4A7C15ULL * (i + 1) ^ (i << 40);
  uint64_t *d;
  hipMalloc(&d, words * 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipMemcpy(d, h, words * 8, hipMemcpyHostToDevice);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_single, dim3((nperm + 63) / 64), dim3(64), 0, 0, d, chain, nperm);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  hipEventElapsedTime(&ms, a, b);
  hipMemcpy(h1, d, words * 8, hipMemcpyDeviceToHost);
  printf("%d permutations, one lane each (fixed registers, %d waves): %.2f us per dependent permutation\n", nperm, (nperm + 63) / 64,
         ms * 1e3 / chain);
  for (int rep = 0; rep < 2; rep++) {
    hipMemcpy(d, h, words * 8, hipMemcpyHostToDevice);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(64 * waves), 0, 0, d, chain);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  hipEventElapsedTime(&ms, a, b);
  hipMemcpy(h2, d, words * 8, hipMemcpyDeviceToHost);
  printf("%d permutations, 25 lanes each (%d blocks x %d waves): %.2f us per dependent permutation, results %s\n", nperm, blocks, waves,
         ms * 1e3 / chain, memcmp(h1, h2, words * 8) ? "DIFFER" : "match");
  return 0;
}
