/*
 * oracle/gl.h -- Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic
 * extension F_p[X]/(X^2 - 7).
 *
 * TEST INFRASTRUCTURE ONLY.  This directory is the CPU oracle (checker) for the
 * HIP proving path.  Nothing under oracle/ is linked into, imported by or
 * executed from the product library (libp2gpu.so); only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it.
 *
 * Restates (un-vendored dependency): plonky2_field 0.2.2
 *   GoldilocksField           (field/src/goldilocks_field.rs)
 *   QuadraticExtension<GL>    (field/src/extension/quadratic.rs, W = 7)
 * Anchored in-tree by: plonky2-backend/src/lib.rs:11-14 (F = GoldilocksField,
 * D = 2) and circuit_translation/tests/test_assert_zero.rs:275-285 (modulus pin
 * p = 18446744069414584321).
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */
#define GL_GENERATOR 14293326489335486720ULL /* MULTIPLICATIVE_GROUP_GENERATOR, also coset shift */
#define GL_TWO_ADICITY 32
#define GL_POWER_OF_TWO_GENERATOR 7277203076849721926ULL /* GL_GENERATOR^((p-1)/2^32); w_64 = 8 */
#define GL_EXT_W 7ULL

static inline gl_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

/* (branch-free forms: the operands are random field elements, a data-dependent branch mispredicts every other call --
 * that, not the multiplier, was most of an oracle butterfly's 20 ns) */
static inline gl_t gl_add(gl_t a, gl_t b) {
  uint64_t s = a + b;
  uint64_t over = (uint64_t)(s < a) | (uint64_t)(s >= GL_P);
  return s - (GL_P & (0 - over));
}
static inline gl_t gl_sub(gl_t a, gl_t b) {
  uint64_t d = a - b;
  return d + (GL_P & (0 - (uint64_t)(a < b)));
}
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

/* x mod p for a 128-bit x, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p). */
static inline gl_t gl_reduce128(u128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hh = hi >> 32, hl = hi & GL_EPS;
  uint64_t t0 = lo - hh;
  t0 -= GL_EPS & (0 - (uint64_t)(lo < hh));
  uint64_t t1 = hl * GL_EPS;
  uint64_t t2 = t0 + t1;
  t2 += GL_EPS & (0 - (uint64_t)(t2 < t1));
  return gl_canon(t2);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((u128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
/* from_noncanonical_u64 */
static inline gl_t gl_from_u64(uint64_t x) { return gl_canon(x); }

static inline gl_t gl_pow(gl_t b, uint64_t e) {
  gl_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }

/* primitive 2^k-th root of unity: POWER_OF_TWO_GENERATOR^(2^(32-k)) */
static inline gl_t gl_root_of_unity(unsigned k) {
  gl_t g = GL_POWER_OF_TWO_GENERATOR;
  for (unsigned i = k; i < GL_TWO_ADICITY; i++) g = gl_sqr(g);
  return g;
}

static inline size_t bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}

/* ---- quadratic extension ------------------------------------------------ */
typedef struct {
  gl_t c0, c1;
} ext_t;

static inline ext_t ext_make(gl_t a, gl_t b) {
  ext_t r = {a, b};
  return r;
}
static inline ext_t ext_from(gl_t a) { return ext_make(a, 0); }
static inline ext_t ext_add(ext_t a, ext_t b) { return ext_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
static inline ext_t ext_sub(ext_t a, ext_t b) { return ext_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
static inline ext_t ext_neg(ext_t a) { return ext_make(gl_neg(a.c0), gl_neg(a.c1)); }
static inline ext_t ext_mul(ext_t a, ext_t b) {
  gl_t c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul(GL_EXT_W, gl_mul(a.c1, b.c1)));
  gl_t c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
  return ext_make(c0, c1);
}
static inline ext_t ext_scale(ext_t a, gl_t s) { return ext_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
static inline int ext_eq(ext_t a, ext_t b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline int ext_is_zero(ext_t a) { return a.c0 == 0 && a.c1 == 0; }
static inline ext_t ext_inv(ext_t a) {
  /* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
  gl_t norm = gl_sub(gl_sqr(a.c0), gl_mul(GL_EXT_W, gl_sqr(a.c1)));
  gl_t ni = gl_inv(norm);
  return ext_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
static inline ext_t ext_pow(ext_t b, uint64_t e) {
  ext_t r = ext_from(1);
  while (e) {
    if (e & 1) r = ext_mul(r, b);
    b = ext_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline ext_t ext_from_u64(uint64_t x) { return ext_from(gl_canon(x)); }

#endif
