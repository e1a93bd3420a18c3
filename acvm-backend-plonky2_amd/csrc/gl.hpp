// gl.hpp -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension
// F_p[X]/(X^2 - 7) for host and gfx950 device code.
//
// This is the arithmetic of plonky2_field 0.2.2 GoldilocksField /
// QuadraticExtension (un-vendored dependency of the reference; in-tree anchors:
// plonky2-backend/src/lib.rs:11-14, tests/test_assert_zero.rs:275-285).
//
// CDNA4 has no 64x64 multiplier: a product is built from v_mul_lo/hi_u32 /
// v_mad_u64_u32 pieces by the compiler (__umul64hi), and the reduction uses only
// shifts/adds because 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).  No MFMA anywhere.
// Values stored in memory are always canonical (< p).
#pragma once
#include <cstdint>
#include <cstddef>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define P2_HD __host__ __device__ __forceinline__
#else
#define P2_HD inline
#endif

namespace p2 {

typedef uint64_t gl_t;
constexpr uint64_t GL_P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t GL_EPS = 0xFFFFFFFFULL;
// plonky2's generators (goldilocks_field.rs), not the "7 / 7^((p-1)/2^32)" pair other Goldilocks
// libraries use: pinned by the reference's own proof artefacts (tests/golden/reference_proofs.py)
constexpr uint64_t GL_GEN = 14293326489335486720ULL;  // MULTIPLICATIVE_GROUP_GENERATOR = coset shift
constexpr uint64_t GL_ROOT_2_32 = 7277203076849721926ULL;   // POWER_OF_TWO_GENERATOR = GL_GEN^((p-1)/2^32); w_64 = 8

#ifndef P2_GL_ASM
#define P2_GL_ASM 1  // 1: carry-chain formulations below on the device; 0: what hipcc makes of the portable code
#endif
#if defined(__HIP_DEVICE_COMPILE__) && P2_GL_ASM
#define P2_GL_DEV_ASM 1
#else
#define P2_GL_DEV_ASM 0
#endif

// ---- portable forms (host; device reference for the self-test) -------------------------------------------
P2_HD gl_t gl_canon_c(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
P2_HD gl_t gl_add_c(gl_t a, gl_t b) {
  uint64_t s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
}
P2_HD gl_t gl_sub_c(gl_t a, gl_t b) { return a >= b ? a - b : a - b + GL_P; }
// reduce hi:lo (a 128-bit value) mod p -> canonical
P2_HD gl_t gl_reduce128_c(uint64_t lo, uint64_t hi) {
  uint64_t hh = hi >> 32, hl = hi & GL_EPS;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= GL_EPS;
  uint64_t t1 = hl * GL_EPS;  // (hl << 32) - hl
  uint64_t t2 = t0 + t1;
  if (t2 < t1) t2 += GL_EPS;
  return gl_canon_c(t2);
}

// wait states inside the carry chains (experiment knobs, scratch/ubench/hazard.hip): P2_HZ_A_N between a VALU that
// leaves a carry in an SGPR pair and the scalar instruction that reads it, P2_HZ_B_N between that scalar
// instruction and the VALU that takes its result as carry-in; 0 = none, 1 = s_nop 0, 2 = s_nop 1
#ifndef P2_HZ_A_N
#define P2_HZ_A_N 0
#endif
#ifndef P2_HZ_B_N
#define P2_HZ_B_N 0
#endif
#if P2_HZ_A_N == 0
#define P2_HZ_A ""
#elif P2_HZ_A_N == 1
#define P2_HZ_A "s_nop 0\n\t"
#else
#define P2_HZ_A "s_nop 1\n\t"
#endif
#if P2_HZ_B_N == 0
#define P2_HZ_B ""
#elif P2_HZ_B_N == 1
#define P2_HZ_B "s_nop 0\n\t"
#else
#define P2_HZ_B "s_nop 1\n\t"
#endif
#if P2_GL_DEV_ASM
// ---- gfx950 forms ------------------------------------------------------------------------------------------
// hipcc lowers the portable code to v_lshl_add_u64 / v_cmp_lt_u64 / v_cndmask_b32 triples: 6 VALU per add or
// sub, 17 per reduction.  Here every wrap of 2^64 (= eps = 2^32 - 1 mod p) is taken from the carry flag the
// 32-bit add already produced and applied as "x0 -/+= m ; x1 +/-= m & !k" (2 VALU + one s_andn2 on the scalar
// unit, which does not compete for the vector issue slot): sub 4, add 5, reduction 8 + 3, canonical out.
// Operand contract: canonical in, canonical out -- and gl_add(a, b) / gl_sub(a, b) accept ANY u64 as their FIRST operand a
// as long as b is canonical: the result is then a u64 CONGRUENT to a +- b (a + b < 2^64 + p wraps at most once and the one
// correction brings it back below 2^64; a >= p > b never borrows), canonical whenever a is.  The gate evaluator relies on
// that for congruent words (gl_mul_nc, value_out ...) in the first position, whose consumers again take any u64; b must be
// < p.  Every other form takes canonical operands unless its comment says otherwise.  p2gpu_field_selftest slots 8-13
// exercise exactly this with operands in [p, 2^64).  Carry chains through vcc run back to back like the ones of gl_mul128; a carry an instruction
// leaves in another SGPR pair is consumed by the scalar unit or at least two instructions later.  The scalar
// mask instructions write SCC: it is in every clobber list (the compiler keeps compares live across the block).
P2_HD gl_t gl_join(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
// x >= p ? x - p : x for any u64: x >= p <=> x1 = 2^32 - 1 and x0 >= 1, and then x - p = (0, x0 - 1)
P2_HD gl_t gl_canon(uint64_t x) {
  const uint64_t g = __builtin_amdgcn_uicmpl(x, GL_P - 1, 34 /* ICMP_UGT */);
  uint32_t r0, r1;
  uint64_t junk;
  asm("s_nop 1\n\t"  // g comes straight from a v_cmp: two wait states before a VALU reads it
      "v_subb_co_u32 %0, %2, %3, 0, %5\n\t"
      "v_addc_co_u32 %1, %2, %4, 0, %5"
      : "=&v"(r0), "=&v"(r1), "=&s"(junk)
      : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "s"(g));
  return gl_join(r0, r1);
}
P2_HD gl_t gl_sub(gl_t a, gl_t b) {
  uint32_t d0, d1;
  uint64_t k;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"   // borrow: the difference wrapped, d + 2^64 = d + p + eps
      "v_addc_co_u32 %0, %2, %0, 0, vcc\n\t"     // d -= eps: d0 += 1 (carry k) ...
      P2_HZ_A "s_andn2_b64 vcc, vcc, %2\n\t" P2_HZ_B
      "v_subb_co_u32 %1, vcc, %1, 0, vcc"          // ... d1 -= 1 - k
      : "=&v"(d0), "=&v"(d1), "=&s"(k)
      : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32))
      : "vcc", "scc");
  return gl_join(d0, d1);
}
P2_HD gl_t gl_add(gl_t a, gl_t b) {
  uint32_t s0, s1;
  uint64_t c;
  asm("v_add_co_u32 %0, vcc, %3, %5\n\t"
      "v_addc_co_u32 %1, %2, %4, %6, vcc"
      : "=&v"(s0), "=&v"(s1), "=&s"(c)
      : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32))
      : "vcc");
  // a + b >= p <=> the 64-bit sum wrapped or is >= p; then subtract p = add eps (mod 2^64)
  const uint64_t m = c | __builtin_amdgcn_uicmpl(gl_join(s0, s1), GL_P - 1, 34);
  uint32_t r0, r1;
  uint64_t k;
  asm("v_subb_co_u32 %0, %2, %3, 0, %5\n\t"      // s0 -= 1 (borrow k) ...
      P2_HZ_A "s_andn2_b64 %2, %5, %2\n\t" P2_HZ_B
      "v_addc_co_u32 %1, %2, %4, 0, %2"            // ... s1 += 1 - k
      : "=&v"(r0), "=&v"(r1), "=&s"(k)
      : "v"(s0), "v"(s1), "s"(m)
      : "scc");
  return gl_join(r0, r1);
}
// (w3 w2 w1 w0) mod p = (w1:w0) + w2 * eps - w3  (2^64 = eps, 2^96 = -1).  With c = carry(w1 + w2):
// ((w1 + w2 + c) : w0) - (w2 + w3 + c), one wrap at most, then the canonical representative.
// (_nc: SOME u64 congruent to the value, not necessarily < p -- for intermediates whose only consumers are products and
// unreduced dot products, which take any u64: three instructions less than the canonical form)
P2_HD uint64_t gl_reduce_words_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  uint32_t r0, r1, y0;
  uint64_t cy, k;
  asm("v_add_co_u32 %1, vcc, %6, %7\n\t"          // u1 = w1 + w2, carry c
      "v_addc_co_u32 %2, %3, %7, %8, vcc\n\t"     // y = w2 + w3 + c (33 bits: carry cy)
      "v_addc_co_u32 %1, vcc, %1, 0, vcc\n\t"     // u1 += c (cannot carry)
      "v_sub_co_u32 %0, vcc, %5, %2\n\t"          // r0 = w0 - y0
      "v_subb_co_u32 %1, vcc, %1, 0, vcc\n\t"     // r1 = u1 - borrow
      "v_subb_co_u32 %1, %4, %1, 0, %3\n\t"       // r1 -= cy
      P2_HZ_A "s_or_b64 vcc, vcc, %4\n\t" P2_HZ_B              // wrapped below zero (the two borrows exclude each other)
      "v_addc_co_u32 %0, %4, %0, 0, vcc\n\t"      // r -= eps
      P2_HZ_A "s_andn2_b64 vcc, vcc, %4\n\t" P2_HZ_B
      "v_subb_co_u32 %1, vcc, %1, 0, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(y0), "=&s"(cy), "=&s"(k)
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3)
      : "vcc", "scc");
  return gl_join(r0, r1);
}
P2_HD gl_t gl_reduce_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return gl_canon(gl_reduce_words_nc(w0, w1, w2, w3)); }
P2_HD gl_t gl_reduce128(uint64_t lo, uint64_t hi) {
  return gl_reduce_words((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
// (w1:w0) + w2 * eps
P2_HD gl_t gl_reduce_add_eps(uint32_t w0, uint32_t w1, uint32_t w2) {
  uint32_t r0, r1, t0, t1;
  uint64_t k;
  asm("v_sub_co_u32 %2, vcc, 0, %7\n\t"           // w2 * eps = (w2 - [w2 != 0]) : -w2
      "v_subbrev_co_u32 %3, vcc, 0, %7, vcc\n\t"
      "v_add_co_u32 %0, vcc, %5, %2\n\t"
      "v_addc_co_u32 %1, vcc, %6, %3, vcc\n\t"    // carry: += eps
      "v_subb_co_u32 %0, %4, %0, 0, vcc\n\t"
      P2_HZ_A "s_andn2_b64 vcc, vcc, %4\n\t" P2_HZ_B
      "v_addc_co_u32 %1, vcc, %1, 0, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&s"(k)
      : "v"(w0), "v"(w1), "v"(w2)
      : "vcc", "scc");
  return gl_canon(gl_join(r0, r1));
}
// z * eps - (h1:h0)
P2_HD gl_t gl_reduce_eps_sub(uint32_t z, uint32_t h0, uint32_t h1) {
  uint32_t r0, r1, t0, t1;
  uint64_t k;
  asm("v_sub_co_u32 %2, vcc, 0, %7\n\t"
      "v_subbrev_co_u32 %3, vcc, 0, %7, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %2, %5\n\t"
      "v_subb_co_u32 %1, vcc, %3, %6, vcc\n\t"    // borrow: -= eps
      "v_addc_co_u32 %0, %4, %0, 0, vcc\n\t"
      P2_HZ_A "s_andn2_b64 vcc, vcc, %4\n\t" P2_HZ_B
      "v_subb_co_u32 %1, vcc, %1, 0, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&s"(k)
      : "v"(h0), "v"(h1), "v"(z)
      : "vcc", "scc");
  return gl_canon(gl_join(r0, r1));
}
#else
P2_HD gl_t gl_canon(uint64_t x) { return gl_canon_c(x); }
P2_HD gl_t gl_add(gl_t a, gl_t b) { return gl_add_c(a, b); }
P2_HD gl_t gl_sub(gl_t a, gl_t b) { return gl_sub_c(a, b); }
P2_HD gl_t gl_reduce128(uint64_t lo, uint64_t hi) { return gl_reduce128_c(lo, hi); }
#endif
#if defined(__HIPCC__)
// A load the scalar unit may always take: hipcc turns a uniform load into s_load only while no store of the kernel can have
// touched the memory before it, so ONE global store inside a loop makes every uniform table look-up of the later iterations a
// vector load that waits out an L2 round trip (round 6: the alpha powers of the gate kernels, the column classes / scalars of
// the leaf hash).  Through the constant address space the load is scalar whatever the kernel stores -- for tables the kernel
// itself never writes.  (A non-uniform address still gives correct code: a vector load.)
template <class T>
__device__ __forceinline__ T ld_uniform(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const T __attribute__((address_space(4))) *cptr;
  return *(cptr)(uintptr_t)p;
#else
  return *p;
#endif
}
#endif
P2_HD gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }
P2_HD gl_t gl_dbl(gl_t a) { return gl_add(a, a); }

#if defined(__HIP_DEVICE_COMPILE__)
// 64 x 64 -> 128 on gfx950 in 8 VALU: four v_mad_u64_u32 (the cross terms chained through the
// carry-out SGPR pair) and one 32-bit carry chain.  hipcc's own lowering of the same product
// needs 5 extra v_mov to build {x, 0} addend pairs; measured 57 vs 62-64 lane-cycles per modmul
// (scratch/ubench/ub3.hip).
__device__ __forceinline__ void gl_mul128(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t p0, m, p3, cdummy, carry;
  uint32_t r1, r2, r3;
  asm("v_mad_u64_u32 %0, %3, %5, %6, 0\n\t"
      "v_mad_u64_u32 %1, %3, %5, %8, 0\n\t"
      "v_mad_u64_u32 %2, %3, %7, %8, 0\n\t"
      "v_mad_u64_u32 %1, %4, %7, %6, %1"
      : "=&v"(p0), "=&v"(m), "=&v"(p3), "=&s"(cdummy), "=&s"(carry)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  asm("v_add_co_u32 %0, vcc, %3, %4\n\t"
      "v_addc_co_u32 %1, vcc, %5, %6, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %7, 0, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, 0, %8"
      : "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"((uint32_t)(p0 >> 32)), "v"((uint32_t)m), "v"((uint32_t)p3), "v"((uint32_t)(m >> 32)),
        "v"((uint32_t)(p3 >> 32)), "s"(carry)
      : "vcc");
  lo = ((uint64_t)r1 << 32) | (uint32_t)p0;
  hi = ((uint64_t)r3 << 32) | r2;
}
#endif
P2_HD gl_t gl_mul(gl_t a, gl_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t lo, hi;
  gl_mul128(a, b, lo, hi);
#else
  unsigned __int128 x = (unsigned __int128)a * b;
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
#endif
  return gl_reduce128(lo, hi);
}
P2_HD gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
#if defined(__HIP_DEVICE_COMPILE__) && P2_GL_DEV_ASM
// a * b for ANY u64 a, b (congruent operands in, congruent u64 out, not canonical)
__device__ __forceinline__ uint64_t gl_mul_nc(uint64_t a, uint64_t b) {
  uint64_t lo, hi;
  gl_mul128(a, b, lo, hi);
  return gl_reduce_words_nc((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ uint64_t gl_reduce128_nc(uint64_t lo, uint64_t hi) {
  return gl_reduce_words_nc((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
#else
P2_HD uint64_t gl_mul_nc(uint64_t a, uint64_t b) { return gl_mul(gl_canon(a), gl_canon(b)); }
P2_HD uint64_t gl_reduce128_nc(uint64_t lo, uint64_t hi) { return gl_reduce128(lo, hi); }
#endif
// a * b + c with one reduction: (p-1)^2 + p < 2^128, so the sum is formed unreduced
P2_HD gl_t gl_mul_add(gl_t a, gl_t b, gl_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t lo, hi;
  gl_mul128(a, b, lo, hi);
  lo += c;
  hi += lo < c;
#else
  unsigned __int128 x = (unsigned __int128)a * b + c;
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
#endif
  return gl_reduce128(lo, hi);
}
// a * b + c as SOME congruent u64 (a, b any u64 with a * b + c < 2^128: e.g. one of them canonical): for factors of a running
// product, whose only consumer is the next multiplication
P2_HD uint64_t gl_mul_add_nc(uint64_t a, uint64_t b, gl_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t lo, hi;
  gl_mul128(a, b, lo, hi);
  lo += c;
  hi += lo < c;
  return gl_reduce128_nc(lo, hi);
#else
  return gl_mul_add(gl_canon(a), gl_canon(b), c);
#endif
}
// multiply by a small constant (< 2^32): the product fits in 96 bits
P2_HD gl_t gl_mul_small(gl_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t lo = a * (uint64_t)k;
  uint64_t hi = __umul64hi(a, (uint64_t)k);
#else
  unsigned __int128 x = (unsigned __int128)a * k;
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
#endif
  // hi < 2^32: only the hl term
  uint64_t t1 = hi * GL_EPS;
  uint64_t t2 = lo + t1;
  if (t2 < t1) t2 += GL_EPS;
  return gl_canon(t2);
}
// Sum of products accumulated UNREDUCED in 160 bits (two 64-bit words + an overflow count) and
// reduced once: a 64x64 product + 5-instruction carry chain per term instead of product +
// Goldilocks reduction + modular add (13 vs 29 VALU).  Good for < 2^31 terms.  Device code only.
#if defined(__HIPCC__)
struct Acc160 {
  uint32_t w[5];
  __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = w[4] = 0; }
  __device__ __forceinline__ void mac(gl_t a, gl_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t lo, hi;
    gl_mul128(a, c, lo, hi);
    asm("v_add_co_u32 %0, vcc, %0, %5\n\t"
        "v_addc_co_u32 %1, vcc, %1, %6, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %7, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc"
        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4])
        : "v"((uint32_t)lo), "v"((uint32_t)(lo >> 32)), "v"((uint32_t)hi), "v"((uint32_t)(hi >> 32))
        : "vcc");
#else
    (void)a; (void)c;  // device-only (the host pass of hipcc only needs this to parse)
#endif
  }
  // lo + 2^64 hi + 2^128 ov  (mod p), 2^128 = -2^32
  __device__ __forceinline__ gl_t value() const {
    const gl_t r = gl_reduce128(((uint64_t)w[1] << 32) | w[0], ((uint64_t)w[3] << 32) | w[2]);
    return gl_sub(r, (uint64_t)w[4] << 32);  // < 2^31 terms, so w[4] << 32 < p
  }
};
#endif
P2_HD gl_t gl_pow(gl_t b, uint64_t e) {
  gl_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
// a^(p-2), p - 2 = 2^64 - 2^32 - 1 = 31 ones, a zero, 32 ones: a^(2^31 - 1) by doubling the run of ones
// (30 squarings + 8 products), then (a^(2^31-1))^(2^33) * a^(2^32-1): 64 squarings + 10 products instead of the
// 63 + 62 of plain square-and-multiply.  gl_inv(0) = 0.
P2_HD gl_t gl_sqr_n(gl_t a, int k) {
  for (int i = 0; i < k; i++) a = gl_mul(a, a);
  return a;
}
P2_HD gl_t gl_inv(gl_t a) {
  const gl_t e2 = gl_mul(gl_mul(a, a), a);            // 2 ones
  const gl_t e4 = gl_mul(gl_sqr_n(e2, 2), e2);
  const gl_t e8 = gl_mul(gl_sqr_n(e4, 4), e4);
  const gl_t e16 = gl_mul(gl_sqr_n(e8, 8), e8);
  const gl_t e24 = gl_mul(gl_sqr_n(e16, 8), e8);
  const gl_t e28 = gl_mul(gl_sqr_n(e24, 4), e4);
  const gl_t e30 = gl_mul(gl_sqr_n(e28, 2), e2);
  const gl_t e31 = gl_mul(gl_mul(e30, e30), a);        // a^(2^31 - 1)
  const gl_t e32 = gl_mul(gl_mul(e31, e31), a);        // a^(2^32 - 1)
  return gl_mul(gl_sqr_n(e31, 33), e32);
}
// primitive 2^k-th root of unity
P2_HD gl_t gl_root(unsigned k) {
  gl_t g = GL_ROOT_2_32;
  for (unsigned i = k; i < 32; i++) g = gl_sqr(g);
  return g;
}
P2_HD uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  uint32_t r = 0;
  for (unsigned i = 0; i < bits; i++) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
#endif
}

struct ext_t {
  gl_t c0, c1;
};
P2_HD ext_t ext_make(gl_t a, gl_t b) {
  ext_t r;
  r.c0 = a;
  r.c1 = b;
  return r;
}
P2_HD ext_t ext_from(gl_t a) { return ext_make(a, 0); }
P2_HD ext_t ext_add(ext_t a, ext_t b) { return ext_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
P2_HD ext_t ext_sub(ext_t a, ext_t b) { return ext_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
P2_HD ext_t ext_mul(ext_t a, ext_t b) {
  gl_t c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul_small(gl_mul(a.c1, b.c1), 7));
  gl_t c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
  return ext_make(c0, c1);
}
P2_HD ext_t ext_scale(ext_t a, gl_t s) { return ext_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
P2_HD bool ext_eq(ext_t a, ext_t b) { return a.c0 == b.c0 && a.c1 == b.c1; }
P2_HD ext_t ext_inv(ext_t a) {
  gl_t norm = gl_sub(gl_sqr(a.c0), gl_mul_small(gl_sqr(a.c1), 7));
  gl_t ni = gl_inv(norm);
  return ext_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
P2_HD ext_t ext_pow(ext_t b, uint64_t e) {
  ext_t r = ext_from(1);
  while (e) {
    if (e & 1) r = ext_mul(r, b);
    b = ext_mul(b, b);
    e >>= 1;
  }
  return r;
}

}  // namespace p2
