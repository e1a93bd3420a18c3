#include <hip/hip_runtime.h>
#include <cstdint>
typedef uint64_t gl_t;
#define EPS 0xFFFFFFFFull
#define PP 0xFFFFFFFF00000001ull
__device__ __forceinline__ gl_t mulA(gl_t a, gl_t b) {
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t p0 = (uint64_t)a0 * b0;
  uint64_t p1 = (uint64_t)a0 * b1 + (p0 >> 32);
  uint64_t p2 = (uint64_t)a1 * b0 + (uint32_t)p1;
  uint64_t hi = (uint64_t)a1 * b1 + (p1 >> 32) + (p2 >> 32);
  uint64_t lo = (p2 << 32) | (uint32_t)p0;
  uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= EPS;
  uint64_t t1 = (uint64_t)hl * 0xFFFFFFFFu;
  uint64_t t2 = t0 + t1;
  if (t2 < t1) t2 += EPS;
  uint64_t t3 = t2 + EPS;
  return t3 < t2 ? t3 : t2;   // t2 >= p  <=>  t2 + EPS overflows
}
// variant B: 32-bit carry chains via builtins
__device__ __forceinline__ gl_t mulB(gl_t a, gl_t b) {
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t p0 = (uint64_t)a0 * b0;
  uint64_t p1 = (uint64_t)a0 * b1 + (p0 >> 32);
  uint64_t p2 = (uint64_t)a1 * b0 + (uint32_t)p1;
  uint64_t hi = (uint64_t)a1 * b1 + (p1 >> 32) + (p2 >> 32);
  uint32_t l0 = (uint32_t)p0, l1 = (uint32_t)p2;
  uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
  // x = lo + (hl << 32) - hl - hh  (mod p), tracking 2^64 wraps as multiples of EPS
  uint32_t x1; bool c1 = __builtin_add_overflow(l1, hl, &x1);       // + hl<<32
  uint64_t x = ((uint64_t)x1 << 32) | l0;
  uint64_t y = (uint64_t)hl + hh;                                    // < 2^33
  uint64_t z = x - y; bool bw = x < y;
  // net wraps: +c1 (2^64 = EPS) and -bw (borrowed 2^64 -> subtract EPS)
  if (c1 && !bw) { uint64_t w = z + EPS; z = (w < z) ? w + EPS : w; }
  else if (!c1 && bw) { uint64_t w = z - EPS; z = (z < EPS) ? w - EPS : w; }
  uint64_t t3 = z + EPS;
  return t3 < z ? t3 : z;
}
__global__ void kA(gl_t *o, const gl_t *a, const gl_t *b) { int i = threadIdx.x; o[i] = mulA(a[i], b[i]); }
__global__ void kB(gl_t *o, const gl_t *a, const gl_t *b) { int i = threadIdx.x; o[i] = mulB(a[i], b[i]); }
