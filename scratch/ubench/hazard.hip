// hazard.hip -- do the carry-chain forms of csrc/gl.hpp need wait states on gfx950?
// The same primitives, one wave alone on its SIMD (instructions issue back to back) and a full grid
// (other waves' instructions in between), against the portable code.  Built several times with different
// P2_HZ_A / P2_HZ_B (s_nop between a VALU's SGPR carry-out and the SALU that reads it / between the SALU
// and the VALU that takes the mask as carry-in); prints the mismatch counters per variant.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../acvm-backend-plonky2_amd/csrc/gl.hpp"
using namespace p2;

__global__ void k(const uint64_t *a, const uint64_t *b, uint32_t n, unsigned long long *bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t x = a[i], y = b[i];
  const gl_t xc = gl_canon_c(x), yc = gl_canon_c(y);
  if (gl_canon(x) != xc) atomicAdd(&bad[0], 1ULL);
  if (gl_add(xc, yc) != gl_add_c(xc, yc)) atomicAdd(&bad[1], 1ULL);
  if (gl_sub(xc, yc) != gl_sub_c(xc, yc)) atomicAdd(&bad[2], 1ULL);
  if (gl_reduce128(x, y) != gl_reduce128_c(x, y)) atomicAdd(&bad[3], 1ULL);
  const uint64_t plo = xc * yc, phi = __umul64hi(xc, yc);
  if (gl_mul(xc, yc) != gl_reduce128_c(plo, phi)) atomicAdd(&bad[4], 1ULL);
#if defined(__HIP_DEVICE_COMPILE__)
  if (gl_reduce_add_eps((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y >> 1) !=
      gl_reduce128_c(x, (uint32_t)y >> 1)) atomicAdd(&bad[5], 1ULL);
  {
    // z * eps - h
    const uint32_t z = (uint32_t)y;
    const gl_t want = gl_sub_c(gl_reduce128_c(0, z), xc);
    if (gl_reduce_eps_sub(z, (uint32_t)xc, (uint32_t)(xc >> 32)) != want) atomicAdd(&bad[6], 1ULL);
  }
#endif
}

int main() {
  const uint64_t P = GL_P, eps = GL_EPS;
  std::vector<uint64_t> pts = {0, 1, 2, eps - 1, eps, eps + 1, eps + 2, 1ULL << 33, (1ULL << 63) - 1, 1ULL << 63,
                               P - eps - 1, P - eps, P - eps + 1, P - 2, P - 1, P, P + 1, P + eps - 2, P + eps - 1,
                               ~0ULL - 1, ~0ULL, 0xFFFFFFFE00000000ULL, 0xFFFFFFFEFFFFFFFFULL, 0x100000000ULL,
                               0x8000000080000000ULL, 0x7FFFFFFF7FFFFFFFULL};
  std::vector<uint64_t> a, b;
  for (auto x : pts) for (auto y : pts) { a.push_back(x); b.push_back(y); }
  uint64_t s = 88172645463325252ULL;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < (1 << 20); i++) {
    uint64_t x = rnd(), y = rnd();
    if (i % 3 == 0) x &= 0xFFFFFFFF00000000ULL;
    if (i % 3 == 1) y |= 0xFFFFFFFF00000000ULL;
    if (i % 5 == 2) x |= 0xFFFFFFFFULL;
    a.push_back(x); b.push_back(y);
  }
  const uint32_t n = (uint32_t)a.size();
  uint64_t *da, *db; unsigned long long *dbad;
  hipMalloc(&da, 8 * n); hipMalloc(&db, 8 * n); hipMalloc(&dbad, 64);
  hipMemcpy(da, a.data(), 8 * n, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 8 * n, hipMemcpyHostToDevice);
  unsigned long long bad[8];
  // (1) one wave at a time: every 64-element slice of the edge pairs in its own launch
  hipMemset(dbad, 0, 64);
  const uint32_t nedge = (uint32_t)(pts.size() * pts.size());
  for (uint32_t off = 0; off < nedge + 4096; off += 64)
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da + off, db + off, 64u, dbad);
  hipMemcpy(bad, dbad, 64, hipMemcpyDeviceToHost);
  printf("single wave : canon %llu add %llu sub %llu reduce %llu mul %llu add_eps %llu eps_sub %llu\n", bad[0], bad[1], bad[2],
         bad[3], bad[4], bad[5], bad[6]);
  // (2) full grid
  hipMemset(dbad, 0, 64);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, n, dbad);
  hipMemcpy(bad, dbad, 64, hipMemcpyDeviceToHost);
  printf("full grid   : canon %llu add %llu sub %llu reduce %llu mul %llu add_eps %llu eps_sub %llu\n", bad[0], bad[1], bad[2],
         bad[3], bad[4], bad[5], bad[6]);
  return 0;
}
