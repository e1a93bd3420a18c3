// gates.hpp -- unfiltered gate-constraint evaluation, generic over the arithmetic:
//   BaseOps: Goldilocks elements -- the quotient kernel (one lane per LDE row), the
//            reference's eval_unfiltered_base_batch path;
//   ExtOps:  quadratic-extension elements on the host -- the prover's self-check of the
//            plonk identity at zeta (the verifier's eval_unfiltered path).
// Constraint ORDER follows the reference exactly (it fixes the alpha powers).
//
// Restated from the reference's in-tree custom gates:
//   plonky2-backend/src/plonky2_ecdsa/biguint/gates/arithmetic_u32.rs:289-348
//   .../add_many_u32.rs:151-192, subtraction_u32.rs:234-270,
//   .../range_check_u32.rs:95-117, comparison.rs:337-414
// and from the stock plonky2 0.2.2 gates the translator emits:
// Noop, Constant, PublicInput, Arithmetic, BaseSum<B>, RandomAccess, Poseidon.
#pragma once
#include "gl.hpp"
#include "poseidon.hpp"

namespace p2 {

enum {
  G_NOOP = 0, G_CONSTANT = 1, G_PUBLIC_INPUT = 2, G_ARITHMETIC = 3, G_BASE_SUM = 4, G_RANDOM_ACCESS = 5,
  G_POSEIDON = 6, G_U32_ARITHMETIC = 7, G_U32_ADD_MANY = 8, G_U32_SUBTRACTION = 9, G_U32_RANGE_CHECK = 10,
  G_COMPARISON = 11, G_KIND_COUNT
};

// unroll factor of the 2-bit limb loops of the U32 gates (each iteration: one wire load, a range check, a Horner step): the
// loads of an unrolled body are issued together -- the memory-level parallelism of a wave in the gate kernels
#ifndef P2_LIMB_UNROLL
#define P2_LIMB_UNROLL 8
#endif
#define P2_PRAGMA_(x) _Pragma(#x)
#define P2_PRAGMA(x) P2_PRAGMA_(x)

struct GateDesc {
  uint32_t kind, p[4];
  uint32_t sel_index, group_start, group_end;
  uint32_t num_constraints, degree, num_constants, pad;  // pad: evaluation group (quotient kernel, heavy mixes)
};

// GateDesc.pad, set at circuit create (handle.hip): which wave of a four-wave block evaluates the gate, and whether its folded
// constraint sum comes from the half-domain evaluation (plonk.hip gate_sums_kernel)
//   bits 0-3   group when every gate is evaluated directly on every row
//   bits 4-7   group in the main kernel when the half-domain gates are looked up there
//   bits 8-11  group in gate_sums_kernel
//   bits 16-23 half-domain slot + 1 (0: evaluated directly)
P2_HD uint32_t gate_group(const GateDesc &g, uint32_t use_half) { return use_half ? (g.pad >> 4) & 15u : g.pad & 15u; }
P2_HD uint32_t gate_sums_group(const GateDesc &g) { return (g.pad >> 8) & 15u; }
P2_HD uint32_t gate_half_slot(const GateDesc &g) { return (g.pad >> 16) & 255u; }

struct BaseOps {
  typedef gl_t T;
  static P2_HD T from(uint64_t x) { return x; }  // x < p
  static P2_HD T add(T a, T b) { return gl_add(a, b); }
  static P2_HD T sub(T a, T b) { return gl_sub(a, b); }
  static P2_HD T mul(T a, T b) { return gl_mul(a, b); }
  // a product / a seventh power whose ONLY consumers take any u64 congruent to it (Consumer::emit's multiply-accumulate, the
  // SmallDot rows of the Poseidon linear layer): the canonicalisation is skipped (gl.hpp _nc)
  static P2_HD T mul_out(T a, T b) { return gl_mul_nc(a, b); }
  static P2_HD T pow7_out(T x) {
    const uint64_t x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x2, x);
    return gl_mul_nc(x4, x3);
  }
  static P2_HD T mul_small(T a, uint32_t k) { return gl_mul_small(a, k); }
  static P2_HD T dbl(T a) { return gl_dbl(a); }
  // acc = acc * 2^bits + v, kept unreduced in 128 bits and reduced once: limb recompositions
  // (sum_j limb_j 2^(bits j)) are chains of these.  The total shift must stay <= 63 bits.
  struct Horner {
    uint64_t lo = 0, hi = 0;
    P2_HD void push(T v, uint32_t bits) {
      hi = (hi << bits) | (lo >> (64 - bits));
      lo <<= bits;
      lo += v;
      hi += lo < v;
    }
    P2_HD T value() const { return gl_reduce128(lo, hi); }
  };
  // sum_i v_i k_i with small k_i (sum of the k_i < 2^31: the fused Poseidon layers reach 2^24.2): the two 32-bit halves of v
  // are multiplied and summed separately in 64 bits (one v_mad_u64_u32 each), one reduction at the end
  struct SmallDot {
    uint64_t lo = 0, hi = 0;
    P2_HD void add(T v, uint32_t k) {
      lo += (uint64_t)(uint32_t)v * k;
      hi += (v >> 32) * k;
    }
    P2_HD T value() const {
      const uint64_t l = lo + (hi << 32);
      return gl_reduce128(l, (hi >> 32) + (l < lo));
    }
    // the same sum as SOME congruent u64: for consumers that take one -- the FIRST operand of add / sub (gl_add and gl_sub
    // stay exact when only their first operand is not canonical), pow7_out, SmallDot::add, out.emit
    P2_HD T value_out() const {
      const uint64_t l = lo + (hi << 32);
      return gl_reduce128_nc(l, (hi >> 32) + (l < lo));
    }
  };
};
struct ExtOps {
  typedef ext_t T;
  static P2_HD T from(uint64_t x) { return ext_from(x); }
  static P2_HD T add(T a, T b) { return ext_add(a, b); }
  static P2_HD T sub(T a, T b) { return ext_sub(a, b); }
  static P2_HD T mul(T a, T b) { return ext_mul(a, b); }
  static P2_HD T mul_out(T a, T b) { return ext_mul(a, b); }
  static P2_HD T pow7_out(T x) {
    const T x2 = ext_mul(x, x), x4 = ext_mul(x2, x2), x3 = ext_mul(x2, x);
    return ext_mul(x4, x3);
  }
  static P2_HD T mul_small(T a, uint32_t k) { return ext_make(gl_mul_small(a.c0, k), gl_mul_small(a.c1, k)); }
  static P2_HD T dbl(T a) { return ext_add(a, a); }
  struct Horner {
    T acc = ext_make(0, 0);
    P2_HD void push(T v, uint32_t bits) { acc = add(mul_small(acc, 1u << bits), v); }
    P2_HD T value() const { return acc; }
  };
  struct SmallDot {
    T acc = ext_make(0, 0);
    P2_HD void add(T v, uint32_t k) { acc = ExtOps::add(acc, mul_small(v, k)); }
    P2_HD T value() const { return acc; }
    P2_HD T value_out() const { return acc; }
  };
};

template <class F>
P2_HD typename F::T range4(typename F::T v) {  // v (v-1) (v-2) (v-3) = u (u + 2), u = v (v - 3): two products
  // u as a congruent word (its consumers: the first operand of an addition, a product): 45 VALU instead of the 62 of
  // "v^2 canonical, 3v by two additions, a subtraction" -- 192 of a U32ArithmeticGate's 216 constraints are this
  typename F::T u = F::mul_out(v, F::sub(v, F::from(3)));
  return F::mul_out(u, F::add(u, F::from(2)));  // (every caller hands the result to out.emit)
}
template <class F>
P2_HD typename F::T range_product(typename F::T v, uint32_t base) {
  if (base == 4) return range4<F>(v);
  typename F::T p = v;
  for (uint32_t x = 1; x + 1 < base; x++) p = F::mul(p, F::sub(v, F::from(x)));
  return base > 1 ? F::mul_out(p, F::sub(v, F::from(base - 1))) : p;  // (the last product goes to out.emit only)
}
// RandomAccessGate list fold: the nested multiplexer x + b (y - x), lowest bit innermost (the
// same expression tree as folding pairs level by level), evaluated depth-first with
// compile-time indices so nothing spills to scratch on the device
template <class F, int LVL, class WF>
P2_HD typename F::T ra_fold(WF &W, uint32_t item0, const typename F::T *bv) {
  if constexpr (LVL == 0) {
    return W(item0);
  } else {
    typename F::T x = ra_fold<F, LVL - 1>(W, item0, bv);
    typename F::T y = ra_fold<F, LVL - 1>(W, item0 + (1u << (LVL - 1)), bv);
    return F::add(x, F::mul(bv[LVL - 1], F::sub(y, x)));
  }
}

// PoseidonGate (plonky2 gates/poseidon.rs), on its own so that a kernel can evaluate just this gate.
// prc: the round constants in the poseidon_device_constants form (in a partial round only word 0 has one; round 26
// carries what the others owe) -- the S-box inputs, the only constrained points, are those of the plain form and of
// upstream's fast factorisation, so all 123 constraint values coincide.  The 23 linear layers between the S-boxes of
// round 3 and those of round 26 are applied three at a time (poseidon.hpp POSEIDON_FUSED: M, M P M, M P M P M).
// Wires: in 0..11, out 12..23, swap 24, delta 25..28, full-0 S-box inputs 29.. (rounds 1-3), partial 65.., full-1 87..
template <class F, class WF, class OUT>
P2_HD void eval_poseidon_gate(WF W, const gl_t *prc, OUT &out) {
  typedef typename F::T T;
  T st[12];
  const T swap = W(24);
  out.emit(F::mul(swap, F::sub(swap, F::from(1))));
  for (int i = 0; i < 4; i++) {
    const T l = W(i), r = W(i + 4), dl = W(25 + i);
    out.emit(F::sub(F::mul(swap, F::sub(r, l)), dl));
    st[i] = F::add(l, dl);
    st[i + 4] = F::sub(r, dl);
  }
  for (int i = 8; i < 12; i++) st[i] = W(i);
  // one full round: constants, (constraints against the S-box input wires), S-boxes; `layer`: also the linear layer
  auto full_round = [&](int r, bool layer) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) st[i] = F::add(st[i], F::from(prc[12 * r + i]));
    if (r != 0) {
      const uint32_t base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 0; i < 12; i++) {
        const T sb = W(base + i);
        out.emit(F::sub(st[i], sb));
        st[i] = sb;
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) st[i] = F::pow7_out(st[i]);  // consumed by the SmallDot rows below only
    if (!layer) return;
    T nx[12];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int row = 0; row < 12; row++) {
      typename F::SmallDot acc;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 0; i < 12; i++) acc.add(st[i], POSEIDON_FUSED.m[row][i]);
      nx[row] = acc.value_out();  // the next round adds a constant to it / subtracts a wire from it (first operands), or raises it
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) st[i] = nx[i];
  };
  // the S-box of partial round r on word 0: the wire holds its input
  auto partial_sbox = [&](int r, T x0) {
    const T sb = W(65 + (r - 4));
    out.emit(F::sub(F::add(x0, F::from(prc[12 * r])), sb));
    return F::pow7_out(sb);
  };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r = 0; r < 3; r++) full_round(r, true);
  full_round(3, false);
  // linear layers r0, r0 + 1, r0 + 2 with the partial S-boxes of rounds r0 + 1, r0 + 2 between them, then round r0 + 3's
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r0 = 3; r0 < 24; r0 += 3) {
    typename F::SmallDot u1, u2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) u1.add(st[i], POSEIDON_FUSED.m[0][i]);
    const T s1 = partial_sbox(r0 + 1, u1.value_out());
    u2.add(s1, POSEIDON_FUSED.m[0][0]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) u2.add(st[i], POSEIDON_FUSED.mpm[0][i]);
    const T s2 = partial_sbox(r0 + 2, u2.value_out());
    T nx[12];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int row = 0; row < 12; row++) {
      typename F::SmallDot acc;
      acc.add(s1, POSEIDON_FUSED.mpm[row][0]);
      acc.add(s2, POSEIDON_FUSED.m[row][0]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 0; i < 12; i++) acc.add(st[i], POSEIDON_FUSED.mpmpm[row][i]);
      nx[row] = acc.value_out();
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 1; i < 12; i++) st[i] = nx[i];
    st[0] = partial_sbox(r0 + 3, nx[0]);
  }
  {  // layers 24 and 25, round 25's S-box between them
    typename F::SmallDot u1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) u1.add(st[i], POSEIDON_FUSED.m[0][i]);
    const T s1 = partial_sbox(25, u1.value_out());
    T nx[12];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int row = 0; row < 12; row++) {
      typename F::SmallDot acc;
      acc.add(s1, POSEIDON_FUSED.m[row][0]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 0; i < 12; i++) acc.add(st[i], POSEIDON_FUSED.mpm[row][i]);
      nx[row] = acc.value_out();
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 12; i++) st[i] = nx[i];
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r = 26; r < 30; r++) full_round(r, true);
  for (int i = 0; i < 12; i++) out.emit(F::sub(st[i], W(12 + i)));
}

// W(c): wire column c of this row; LC(i): local constant i; pih: public_inputs_hash;
// prc: the 360 Poseidon round constants, poseidon_device_constants form (only read when POSEIDON); out.emit(c) consumes
// the constraints in order.
template <class F, bool POSEIDON, class WF, class CF, class OUT>
P2_HD void eval_gate(const GateDesc &g, WF W, CF LC, const typename F::T *pih, const gl_t *prc, OUT &out) {
  typedef typename F::T T;
  switch (g.kind) {
  case G_NOOP:
    break;
  case G_CONSTANT:
    for (uint32_t i = 0; i < g.p[0]; i++) out.emit(F::sub(LC(i), W(i)));
    break;
  case G_PUBLIC_INPUT:
    for (uint32_t i = 0; i < 4; i++) out.emit(F::sub(W(i), pih[i]));
    break;
  case G_ARITHMETIC: {
    const T c0 = LC(0), c1 = LC(1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
    for (uint32_t i = 0; i < g.p[0]; i++) {
      T m0 = W(4 * i), m1 = W(4 * i + 1), ad = W(4 * i + 2), o = W(4 * i + 3);
      T comp = F::add(F::mul(F::mul(m0, m1), c0), F::mul(ad, c1));
      out.emit(F::sub(o, comp));
    }
    break;
  }
  case G_BASE_SUM: {
    const uint32_t B = g.p[0], L = g.p[1];
    T acc = F::from(0);
    const uint32_t lb = B == 2 ? 1 : (B == 4 ? 2 : 0);
    if (lb && lb * L <= 63) {
      typename F::Horner h;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
      for (uint32_t i = L; i-- > 0;) h.push(W(1 + i), lb);
      acc = h.value();
    } else {
      for (uint32_t i = L; i-- > 0;) acc = F::add(F::mul_small(acc, B), W(1 + i));
    }
    out.emit(F::sub(acc, W(0)));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
    for (uint32_t i = 0; i < L; i++) out.emit(range_product<F>(W(1 + i), B));
    break;
  }
  case G_RANDOM_ACCESS: {
    const uint32_t bits = g.p[0], copies = g.p[1], extra = g.p[2], vec = 1u << bits;
    const uint32_t routed = (2 + vec) * copies + extra;
    for (uint32_t c = 0; c < copies; c++) {
      const uint32_t base = (2 + vec) * c, bw = routed + c * bits;
      for (uint32_t b = 0; b < bits; b++) {
        T bv = W(bw + b);
        out.emit(F::mul(bv, F::sub(bv, F::from(1))));
      }
      T rec = F::from(0);
      for (uint32_t b = bits; b-- > 0;) rec = F::add(F::dbl(rec), W(bw + b));
      out.emit(F::sub(rec, W(base)));
      T bv[6];
      for (uint32_t b = 0; b < 6; b++) bv[b] = F::from(0);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (uint32_t b = 0; b < 6; b++)
        if (b < bits) bv[b] = W(bw + b);
      T folded;
      switch (bits) {
      case 1: folded = ra_fold<F, 1>(W, base + 2, bv); break;
      case 2: folded = ra_fold<F, 2>(W, base + 2, bv); break;
      case 3: folded = ra_fold<F, 3>(W, base + 2, bv); break;
      case 4: folded = ra_fold<F, 4>(W, base + 2, bv); break;
      case 5: folded = ra_fold<F, 5>(W, base + 2, bv); break;
      case 6: folded = ra_fold<F, 6>(W, base + 2, bv); break;
      default: folded = W(base + 2); break;
      }
      out.emit(F::sub(folded, W(base + 1)));
    }
    for (uint32_t i = 0; i < extra; i++) out.emit(F::sub(LC(i), W((2 + vec) * copies + i)));
    break;
  }
  case G_POSEIDON:
    if constexpr (POSEIDON) eval_poseidon_gate<F>(W, prc, out);
    break;
  case G_U32_ARITHMETIC: {
    const uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      T m0 = W(6 * i), m1 = W(6 * i + 1), ad = W(6 * i + 2);
      T lo = W(6 * i + 3), hi = W(6 * i + 4), inv = W(6 * i + 5);
      T computed = F::add(F::mul(m0, m1), ad);
      T diff = F::sub(F::from(0xFFFFFFFFULL), hi);
      T hi_not_max = F::sub(F::mul(inv, diff), F::from(1));
      out.emit(F::mul(hi_not_max, lo));
      T combined = F::add(F::mul(hi, F::from(1ULL << 32)), lo);
      out.emit(F::sub(combined, computed));
      typename F::Horner cl, ch;
#if defined(__HIP_DEVICE_COMPILE__)
P2_PRAGMA(unroll P2_LIMB_UNROLL)
#endif
      for (uint32_t j = 32; j-- > 0;) {
        T limb = W(6 * ops + 32 * i + j);
        out.emit(range4<F>(limb));
        if (j < 16) cl.push(limb, 2);
        else ch.push(limb, 2);
      }
      out.emit(F::sub(cl.value(), lo));
      out.emit(F::sub(ch.value(), hi));
    }
    break;
  }
  case G_U32_ADD_MANY: {
    const uint32_t na = g.p[0], ops = g.p[1];
    for (uint32_t i = 0; i < ops; i++) {
      const uint32_t b = (na + 3) * i;
      T computed = F::from(0);
      for (uint32_t j = 0; j <= na; j++) computed = F::add(computed, W(b + j));
      T res = W(b + na + 1), oc = W(b + na + 2);
      T combined = F::add(F::mul(oc, F::from(1ULL << 32)), res);
      out.emit(F::sub(combined, computed));
      typename F::Horner cr, cc;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 6
#endif
      for (uint32_t j = 18; j-- > 0;) {
        T limb = W((na + 3) * ops + 18 * i + j);
        out.emit(range4<F>(limb));
        if (j < 16) cr.push(limb, 2);
        else cc.push(limb, 2);
      }
      out.emit(F::sub(cr.value(), res));
      out.emit(F::sub(cc.value(), oc));
    }
    break;
  }
  case G_U32_SUBTRACTION: {
    const uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      T x = W(5 * i), y = W(5 * i + 1), bin = W(5 * i + 2), res = W(5 * i + 3), bout = W(5 * i + 4);
      T init = F::sub(F::sub(x, y), bin);
      out.emit(F::sub(res, F::add(init, F::mul(bout, F::from(1ULL << 32)))));
      typename F::Horner comb;
#if defined(__HIP_DEVICE_COMPILE__)
P2_PRAGMA(unroll P2_LIMB_UNROLL)
#endif
      for (uint32_t j = 16; j-- > 0;) {
        T limb = W(5 * ops + 16 * i + j);
        out.emit(range4<F>(limb));
        comb.push(limb, 2);
      }
      out.emit(F::sub(comb.value(), res));
      out.emit(F::mul(bout, F::sub(F::from(1), bout)));
    }
    break;
  }
  case G_U32_RANGE_CHECK: {
    const uint32_t nl = g.p[0];
    for (uint32_t i = 0; i < nl; i++) {
      typename F::Horner sum;
#if defined(__HIP_DEVICE_COMPILE__)
P2_PRAGMA(unroll P2_LIMB_UNROLL)
#endif
      for (uint32_t j = 16; j-- > 0;) sum.push(W(nl + 16 * i + j), 2);
      out.emit(F::sub(sum.value(), W(i)));
#if defined(__HIP_DEVICE_COMPILE__)
P2_PRAGMA(unroll P2_LIMB_UNROLL)
#endif
      for (uint32_t j = 0; j < 16; j++) out.emit(range4<F>(W(nl + 16 * i + j)));
    }
    break;
  }
  case G_COMPARISON: {
    const uint32_t nb = g.p[0], nc = g.p[1], cb = (nb + nc - 1) / nc;
    const uint32_t fc = 4, sc = 4 + nc, dm = 4 + 2 * nc, eq = 4 + 3 * nc, im = 4 + 4 * nc, msb = 4 + 5 * nc;
    T a = F::from(0), b = F::from(0);
    if (cb * nc <= 63) {
      typename F::Horner ha, hb;
      for (uint32_t i = nc; i-- > 0;) {
        ha.push(W(fc + i), cb);
        hb.push(W(sc + i), cb);
      }
      a = ha.value();
      b = hb.value();
    } else {
      for (uint32_t i = nc; i-- > 0;) {
        a = F::add(F::mul_small(a, 1u << cb), W(fc + i));
        b = F::add(F::mul_small(b, 1u << cb), W(sc + i));
      }
    }
    out.emit(F::sub(a, W(0)));
    out.emit(F::sub(b, W(1)));
    T msd_so_far = F::from(0);
    for (uint32_t i = 0; i < nc; i++) {
      T f = W(fc + i), s = W(sc + i);
      out.emit(range_product<F>(f, 1u << cb));
      out.emit(range_product<F>(s, 1u << cb));
      T diff = F::sub(s, f);
      T e = W(eq + i), iv = W(im + i);
      out.emit(F::sub(F::mul(diff, W(dm + i)), F::sub(F::from(1), e)));
      out.emit(F::mul(e, diff));
      out.emit(F::sub(iv, F::mul(e, msd_so_far)));
      msd_so_far = F::add(iv, F::mul(F::sub(F::from(1), e), diff));
    }
    T msd = W(3);
    out.emit(F::sub(msd, msd_so_far));
    T bits = F::from(0);
    for (uint32_t i = 0; i < cb + 1; i++) {
      T bt = W(msb + i);
      out.emit(F::mul(bt, F::sub(F::from(1), bt)));
    }
    for (uint32_t i = cb + 1; i-- > 0;) bits = F::add(F::dbl(bits), W(msb + i));
    out.emit(F::sub(F::add(msd, F::from(1ULL << cb)), bits));
    out.emit(F::sub(W(2), W(msb + cb)));
    break;
  }
  default:
    break;
  }
}

// gates/selectors.rs compute_filter
template <class F>
P2_HD typename F::T gate_filter(const GateDesc &g, uint32_t gi, uint32_t num_selectors, typename F::T s) {
  typename F::T f = F::from(1);
  for (uint32_t i = g.group_start; i < g.group_end; i++)
    if (i != gi) f = F::mul(f, F::sub(F::from(i), s));
  if (num_selectors > 1) f = F::mul(f, F::sub(F::from(0xFFFFFFFFULL), s));
  return f;
}

}  // namespace p2
