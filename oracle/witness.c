/*
 * oracle/witness.c -- CPU restatement of the per-row witness generators ("N1" of SURVEY.md
 * 8(f)): given a wire matrix whose gate INPUT wires are set, compute every wire a gate's own
 * SimpleGenerator derives from them.  TEST INFRASTRUCTURE ONLY (checker for p2gpu_fill_witness).
 *
 * In-tree generators restated here (plonky2-backend/src/plonky2_ecdsa/biguint/gates/):
 *   U32ArithmeticGenerator  arithmetic_u32.rs:376-426
 *   U32AddManyGenerator     add_many_u32.rs:329-378
 *   U32SubtractionGenerator subtraction_u32.rs:298-343
 *   U32RangeCheckGenerator  range_check_u32.rs:198-220
 *   ComparisonGenerator     comparison.rs:439-537
 * Stock plonky2 0.2.2 generators (absent crate, [P2-recall]): ArithmeticBaseGenerator,
 * BaseSplitGenerator, RandomAccessGenerator, ConstantGate's constant wires, PoseidonGenerator.
 * Copy-constraint propagation between rows (the partition witness) is NOT part of this: it
 * stays with the CPU witness generator; this is the row-local tail of it.
 */
#include "oracle.h"
#include "circuit.h"
#include <stdlib.h>
#include <string.h>

const circuit_t *orc_circuit_inner(const orc_circuit *oc);

static void fill_row(const circuit_t *c, const gate_t *g, size_t row, gl_t *wires, const gl_t *lc) {
  const size_t n = c->n;
#define Wv(col) wires[(size_t)(col) * n + row]
  switch (g->kind) {
  case G_CONSTANT:
    for (uint32_t i = 0; i < g->p[0]; i++) Wv(i) = lc[i];
    break;
  case G_ARITHMETIC:
    for (uint32_t i = 0; i < g->p[0]; i++)
      Wv(4 * i + 3) = gl_add(gl_mul(gl_mul(Wv(4 * i), Wv(4 * i + 1)), lc[0]), gl_mul(Wv(4 * i + 2), lc[1]));
    break;
  case G_BASE_SUM: {
    uint64_t v = Wv(0);
    for (uint32_t i = 0; i < g->p[1]; i++) {
      Wv(1 + i) = v % g->p[0];
      v /= g->p[0];
    }
    break;
  }
  case G_RANDOM_ACCESS: {
    uint32_t bits = g->p[0], copies = g->p[1], extra = g->p[2], vec = 1u << bits;
    uint32_t routed = (2 + vec) * copies + extra;
    for (uint32_t cp = 0; cp < copies; cp++) {
      uint32_t base = (2 + vec) * cp;
      uint64_t idx = Wv(base);
      Wv(base + 1) = Wv(base + 2 + (idx & (vec - 1)));
      for (uint32_t k = 0; k < bits; k++) Wv(routed + cp * bits + k) = (idx >> k) & 1;
    }
    for (uint32_t i = 0; i < extra; i++) Wv((2 + vec) * copies + i) = lc[i];
    break;
  }
  case G_POSEIDON: {
    const gl_t *rc = poseidon_round_constants();
    gl_t st[12], swap = Wv(24);
    for (int i = 0; i < 4; i++) {
      gl_t dl = gl_mul(swap, gl_sub(Wv(i + 4), Wv(i)));
      Wv(25 + i) = dl;
      st[i] = gl_add(Wv(i), dl);
      st[i + 4] = gl_sub(Wv(i + 4), dl);
    }
    for (int i = 8; i < 12; i++) st[i] = Wv(i);
    for (int r = 0; r < 30; r++) {
      for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * r + i]);
      int full = r < 4 || r >= 26;
      if (full && r != 0) {
        uint32_t base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
        for (int i = 0; i < 12; i++) Wv(base + i) = st[i];
      }
      if (!full) Wv(65 + (r - 4)) = st[0];
      for (int i = 0; i < (full ? 12 : 1); i++) {
        gl_t x2 = gl_sqr(st[i]), x4 = gl_sqr(x2), x3 = gl_mul(x2, st[i]);
        st[i] = gl_mul(x4, x3);
      }
      static const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      gl_t nx[12];
      for (int rw = 0; rw < 12; rw++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)st[(i + rw) % 12] * CIRC[i];
        if (rw == 0) acc += (u128)st[0] * 8;
        nx[rw] = gl_reduce128(acc);
      }
      memcpy(st, nx, sizeof nx);
    }
    for (int i = 0; i < 12; i++) Wv(12 + i) = st[i];
    break;
  }
  case G_U32_ARITHMETIC: { /* arithmetic_u32.rs:376-426 */
    uint32_t ops = g->p[0];
    for (uint32_t i = 0; i < ops; i++) {
      uint64_t o = gl_add(gl_mul(Wv(6 * i), Wv(6 * i + 1)), Wv(6 * i + 2));
      uint64_t hi = o >> 32, lo = o & 0xFFFFFFFFull;
      Wv(6 * i + 3) = lo;
      Wv(6 * i + 4) = hi;
      uint64_t diff = 0xFFFFFFFFull - hi;
      Wv(6 * i + 5) = diff ? gl_inv(diff) : 0;
      for (uint32_t j = 0; j < 32; j++) {
        Wv(6 * ops + 32 * i + j) = o & 3;
        o >>= 2;
      }
    }
    break;
  }
  case G_U32_ADD_MANY: { /* add_many_u32.rs:329-378 */
    uint32_t na = g->p[0], ops = g->p[1];
    for (uint32_t i = 0; i < ops; i++) {
      uint32_t b = (na + 3) * i;
      gl_t sum = 0;
      for (uint32_t j = 0; j <= na; j++) sum = gl_add(sum, Wv(b + j));
      uint64_t res = sum & 0xFFFFFFFFull, carry = sum >> 32;
      Wv(b + na + 1) = res;
      Wv(b + na + 2) = carry;
      for (uint32_t j = 0; j < 16; j++) Wv((na + 3) * ops + 18 * i + j) = (res >> (2 * j)) & 3;
      for (uint32_t j = 0; j < 2; j++) Wv((na + 3) * ops + 18 * i + 16 + j) = (carry >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_SUBTRACTION: { /* subtraction_u32.rs:298-343 */
    uint32_t ops = g->p[0];
    for (uint32_t i = 0; i < ops; i++) {
      gl_t init = gl_sub(gl_sub(Wv(5 * i), Wv(5 * i + 1)), Wv(5 * i + 2));
      gl_t bout = init > (1ull << 32) ? 1 : 0;
      gl_t res = gl_add(init, gl_mul(bout, 1ull << 32));
      Wv(5 * i + 3) = res;
      Wv(5 * i + 4) = bout;
      for (uint32_t j = 0; j < 16; j++) Wv(5 * ops + 16 * i + j) = (res >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_RANGE_CHECK: { /* range_check_u32.rs:198-220 (the limb is truncated to u32 there) */
    uint32_t nl = g->p[0];
    for (uint32_t i = 0; i < nl; i++) {
      uint32_t v = (uint32_t)Wv(i);
      for (uint32_t j = 0; j < 16; j++) Wv(nl + 16 * i + j) = (v >> (2 * j)) & 3;
    }
    break;
  }
  case G_COMPARISON: { /* comparison.rs:439-537 */
    uint32_t nb = g->p[0], nc = g->p[1], cb = (nb + nc - 1) / nc;
    uint64_t a = Wv(0), b = Wv(1), cs = 1ull << cb, ta = a, tb = b;
    Wv(2) = a <= b ? 1 : 0;
    gl_t msd = 0;
    for (uint32_t i = 0; i < nc; i++) {
      gl_t f = ta % cs, s = tb % cs;
      ta /= cs;
      tb /= cs;
      Wv(4 + i) = f;
      Wv(4 + nc + i) = s;
      Wv(4 + 2 * nc + i) = (f == s) ? 1 : gl_inv(gl_sub(s, f));
      Wv(4 + 3 * nc + i) = (f == s) ? 1 : 0;
      if (f != s) {
        msd = gl_sub(s, f);
        Wv(4 + 4 * nc + i) = 0;
      } else {
        Wv(4 + 4 * nc + i) = msd;
      }
    }
    Wv(3) = msd;
    uint64_t v = gl_add(cs, msd);
    for (uint32_t i = 0; i < cb + 1; i++) {
      Wv(4 + 5 * nc + i) = v & 1;
      v >>= 1;
    }
    break;
  }
  default:
    break;
  }
#undef Wv
}

int orc_fill_witness(const orc_circuit *oc, uint64_t *wires) {
  const circuit_t *c = orc_circuit_inner(oc);
  poseidon_init();
#pragma omp parallel for schedule(static)
  for (size_t row = 0; row < c->n; row++) {
    /* the row's gate: the one selector column that is not UNUSED holds its index */
    uint32_t gi = 0;
    for (uint32_t s = 0; s < c->num_selectors; s++) {
      gl_t v = c->constants[(size_t)s * c->n + row];
      if (c->num_selectors == 1 || v != 0xFFFFFFFFull) gi = (uint32_t)v;
    }
    if (gi >= c->num_gates) continue;
    gl_t lc[8] = {0};
    for (uint32_t k = 0; k + c->num_selectors < c->num_constants && k < 8; k++)
      lc[k] = c->constants[(size_t)(c->num_selectors + k) * c->n + row];
    fill_row(c, &c->gates[gi], row, wires, lc);
  }
  return 0;
}
