// glphi.hpp -- Goldilocks elements as a + b*phi with phi = 2^32 and SIGNED 64-bit components, for
// the in-register butterfly networks of the NTT (ntt.hip).
//
// p = phi^2 - phi + 1, so phi is a primitive 6th root of unity (phi^3 = -1) and Z[phi]/(phi^2 - phi + 1)
// maps onto F_p.  In this form
//   * an addition / subtraction is two independent 64-bit adds with NO modular correction
//     (2 x v_lshl_add_u64 resp. 2 x (v_sub_co, v_subb)) instead of a canonical gl_add / gl_sub
//     (add, two compares, conditional subtract: 6 VALU each);
//   * a multiplication by 2^(32 q) is a component swap with signs ((a, b) phi = (-b, a + b)); the signs
//     are compile-time knowledge that the next butterfly absorbs (u + (-t) is u - t), so it costs one add;
//   * a multiplication by 2^r, r < 32, is two 64-bit shifts (2 x v_lshlrev_b64) instead of a
//     shift + 96-bit fold + canonicalisation (14-20 VALU): every twiddle INSIDE a radix-16 butterfly
//     network is such a power of two (w_64 = 8 in Goldilocks);
//   * a general twiddle multiplication lands in this form straight from the 128-bit product
//     w0 + w1 phi + w2 phi^2 + w3 phi^3 = (w0 - w2 - w3) + (w1 + w2) phi, without the modular fold.
// The price: components grow (one bit per add, r bits per shift) and must stay inside int64; a
// compile-time schedule (phi_sched below) tracks an exact magnitude bound for both components of every
// register of the network and inserts a renormalisation (phi_norm, 7 VALU: back to ~2^32) exactly
// where the next operation could cross 2^63.
// Leaving the form (phi_to_u64) costs ~13 VALU and yields a u64 in [0, 2^64) that is congruent to the
// value but not necessarily canonical -- which is all the next round's multiplication needs.
//
// Everything is plain C++ on int64_t/uint64_t, so the same code runs on the host
// (tests/test_ntt_phi.py drives csrc/tests/phi_selftest.cpp through every network shape with extreme
// inputs) and on gfx950.  Replaces nothing in the reference by itself: it is the arithmetic under
// plonky2 0.2.2 field/src/fft.rs fft_classic as used by `circuit_data.prove`
// (plonky2-backend/src/actions/prove_action.rs:96), SURVEY.md 8a row P3.
#pragma once
#include "gl.hpp"
#include <type_traits>

namespace p2 {

struct phi_t {
  int64_t a, b;
};

P2_HD int64_t phi_shl(int64_t x, int r) { return (int64_t)((uint64_t)x << r); }

// any u64 (canonical or not): lo + hi * 2^32
P2_HD phi_t phi_from(uint64_t x) {
  phi_t v;
  v.a = (int64_t)(x & 0xFFFFFFFFull);
  v.b = (int64_t)(x >> 32);
  return v;
}

// x * w for any u64 x, w: the 128-bit product folded with phi^2 = phi - 1, phi^3 = -1.
// |a| < 2^33, 0 <= b < 2^33.
P2_HD phi_t phi_mul_u64(uint64_t x, uint64_t w) {
  uint64_t lo, hi;
#if defined(__HIP_DEVICE_COMPILE__)
  gl_mul128(x, w, lo, hi);
#else
  unsigned __int128 pr = (unsigned __int128)x * w;
  lo = (uint64_t)pr;
  hi = (uint64_t)(pr >> 64);
#endif
  const uint64_t w0 = lo & 0xFFFFFFFFull, w1 = lo >> 32, w2 = hi & 0xFFFFFFFFull, w3 = hi >> 32;
  phi_t v;
  v.a = (int64_t)w0 - (int64_t)(w2 + w3);
  v.b = (int64_t)(w1 + w2);
  return v;
}

// same value, components back near 2^32: a = a0 + a1 phi, b = b0 + b1 phi (x0 the low 32 bits as an
// unsigned number, x1 = x >> 32 arithmetic), and a0 + (a1 + b0) phi + b1 phi^2 = (a0 - b1) + (a1 + b0 + b1) phi.
// Valid for every int64 input (|a1|, |b1| <= 2^31); |a'| <= 2^32 - 1 + |b1|, |b'| <= 2^32 - 1 + |a1| + |b1|.
P2_HD phi_t phi_norm(phi_t v) {
  const int64_t a0 = (int64_t)((uint64_t)v.a & 0xFFFFFFFFull), a1 = v.a >> 32;
  const int64_t b0 = (int64_t)((uint64_t)v.b & 0xFFFFFFFFull), b1 = v.b >> 32;
  phi_t r;
  r.a = a0 - b1;
  r.b = a1 + b1 + b0;
  return r;
}

P2_HD phi_t phi_add(phi_t x, phi_t y) {
  phi_t r;
  r.a = x.a + y.a;
  r.b = x.b + y.b;
  return r;
}
P2_HD phi_t phi_sub(phi_t x, phi_t y) {
  phi_t r;
  r.a = x.a - y.a;
  r.b = x.b - y.b;
  return r;
}

// x * 2^E, 0 <= E < 192 (2^192 = 1): E = 32 q + r; the shift grows the components by r bits, phi^q by
// at most one more.  Negations are left to the compiler to fold into the butterfly that follows.
template <int E>
P2_HD phi_t phi_mul_pow2(phi_t x) {
  static_assert(E >= 0 && E < 192, "exponent of 2 out of range");
  constexpr int q = E / 32, r = E % 32;
  const int64_t a = r ? phi_shl(x.a, r) : x.a, b = r ? phi_shl(x.b, r) : x.b;
  phi_t v;
  if constexpr (q == 0) { v.a = a; v.b = b; }
  else if constexpr (q == 1) { v.a = -b; v.b = a + b; }
  else if constexpr (q == 2) { v.a = -(a + b); v.b = a; }
  else if constexpr (q == 3) { v.a = -a; v.b = -b; }
  else if constexpr (q == 4) { v.a = b; v.b = -(a + b); }
  else { v.a = a + b; v.b = -a; }
  return v;
}
// a + b 2^32 (mod p) as a u64 in [0, 2^64) -- congruent, not necessarily < p.
// Needs |b| < 2^63 - 2^34 (then |hi| < 2^31 and |t| < 2^63); any a.
P2_HD uint64_t phi_to_u64(phi_t v) {
  const uint64_t ua = (uint64_t)v.a, ub = (uint64_t)v.b;
  const uint64_t lo = ua + (ub << 32);
  const int64_t c1 = lo < ua;
  const int64_t hi = (v.a >> 63) + (v.b >> 32) + c1;  // V = lo + hi 2^64 exactly, |hi| < 2^31
  const int64_t t = hi * (int64_t)0xFFFFFFFFll;       // 2^64 = 2^32 - 1 (mod p); |t| < 2^63
  uint64_t r = lo + (uint64_t)t;
  const int64_t c2 = r < lo;
  const int64_t m = c2 - (int64_t)(t < 0);            // the true sum is r + m 2^64
  r += (uint64_t)(m * (int64_t)0xFFFFFFFFll);
  return r;
}

// ---- compile-time growth schedule of a 2^LOGR-point butterfly network ------------------------------
// exponent of 2 for the constant twiddle w_{2^(lam+1)}^q (w_64 = 2^3), mod 192
constexpr int phi_tw_exp(int lam, int q, bool inv) {
  const int e = (3 * (32 >> lam) * q) % 192;
  return inv ? (192 - e) % 192 : e;
}
// magnitude bounds |a| <= ma, |b| <= mb (exact integer arithmetic; everything stays <= PHI_MAX)
struct PhiBound {
  uint64_t ma, mb;
};
constexpr uint64_t PHI_MAX = 0x7FFFFFFFFFFFFFFFull - (1ull << 35);  // margin: phi_to_u64 wants |b| < 2^63 - 2^34
constexpr uint64_t PHI_OVER = ~0ull;                                // "does not fit"
constexpr uint64_t phi_sat_add(uint64_t x, uint64_t y) { return (x > PHI_MAX || y > PHI_MAX || x + y > PHI_MAX) ? PHI_OVER : x + y; }
constexpr uint64_t phi_sat_shl(uint64_t x, int r) { return (x > PHI_MAX || (r && (x >> (63 - r)) != 0) || (x << r) > PHI_MAX) ? PHI_OVER : x << r; }
constexpr bool phi_fits(PhiBound v) { return v.ma <= PHI_MAX && v.mb <= PHI_MAX; }
constexpr PhiBound phi_bound_from() { return PhiBound{0xFFFFFFFFull, 0xFFFFFFFFull}; }
constexpr PhiBound phi_bound_mul() { return PhiBound{(1ull << 33) - 2, (1ull << 33) - 2}; }
constexpr PhiBound phi_bound_norm(PhiBound v) {
  const uint64_t a1 = (v.ma >> 32) + 1, b1 = (v.mb >> 32) + 1;
  return PhiBound{0xFFFFFFFFull + b1, 0xFFFFFFFFull + a1 + b1};
}
constexpr PhiBound phi_bound_pow2(PhiBound v, int e) {
  const int q = (e / 32) % 3, r = e % 32;
  const uint64_t a = phi_sat_shl(v.ma, r), b = phi_sat_shl(v.mb, r), ab = phi_sat_add(a, b);
  return q == 0 ? PhiBound{a, b} : (q == 1 ? PhiBound{b, ab} : PhiBound{ab, a});
}
constexpr PhiBound phi_bound_add(PhiBound x, PhiBound y) { return PhiBound{phi_sat_add(x.ma, y.ma), phi_sat_add(x.mb, y.mb)}; }
constexpr PhiBound phi_bound_join(PhiBound x, PhiBound y) { return PhiBound{x.ma > y.ma ? x.ma : y.ma, x.mb > y.mb ? x.mb : y.mb}; }

struct PhiSched {
  bool pre[4][16];  // normalise register i before layer `lam` reads it
  bool mid[4][16];  // DIF: normalise the difference (u - x) before its twiddle shift
  PhiBound out;     // bound over all outputs
  bool ok;          // false: some value cannot be kept inside int64 even with normalisation
};
// DIT: v[j], v[k] = u + t, u - t with t = v[k] 2^e;  DIF: v[j] = u + x, v[k] = (u - x) 2^e.
template <int LOGR, int DIT, bool INV>
constexpr PhiSched phi_sched(PhiBound in) {
  PhiSched s{};
  s.ok = true;
  PhiBound B[16] = {};
  for (int i = 0; i < 16; i++) B[i] = in;
  for (int lc = 0; lc < LOGR; lc++) {
    const int lam = DIT ? lc : LOGR - 1 - lc;
    for (int j = 0; j < (1 << LOGR); j++) {
      if (j & (1 << lam)) continue;
      const int k = j | (1 << lam);
      const int e = phi_tw_exp(lam, j & ((1 << lam) - 1), INV);
      if (DIT) {
        PhiBound t = phi_bound_pow2(B[k], e);
        if (!phi_fits(phi_bound_add(B[j], t))) {  // first try: normalise the twiddled operand
          s.pre[lam][k] = true;
          B[k] = phi_bound_norm(B[k]);
          t = phi_bound_pow2(B[k], e);
        }
        if (!phi_fits(phi_bound_add(B[j], t))) {
          s.pre[lam][j] = true;
          B[j] = phi_bound_norm(B[j]);
        }
        const PhiBound o = phi_bound_add(B[j], t);
        if (!phi_fits(o)) s.ok = false;
        B[j] = o;
        B[k] = o;
      } else {
        if (!phi_fits(phi_bound_add(B[j], B[k]))) {
          s.pre[lam][j] = true;
          s.pre[lam][k] = true;
          B[j] = phi_bound_norm(B[j]);
          B[k] = phi_bound_norm(B[k]);
        }
        PhiBound o = phi_bound_add(B[j], B[k]);
        if (!phi_fits(o)) s.ok = false;
        B[j] = o;
        PhiBound t = phi_bound_pow2(o, e);
        if (!phi_fits(t)) {  // normalise the difference before it is shifted
          s.mid[lam][k] = true;
          o = phi_bound_norm(o);
          t = phi_bound_pow2(o, e);
        }
        if (!phi_fits(t)) s.ok = false;
        B[k] = t;
      }
    }
  }
  PhiBound m = PhiBound{0, 0};
  for (int i = 0; i < (1 << LOGR); i++) m = phi_bound_join(m, B[i]);
  s.out = m;
  return s;
}

// ---- the butterfly network itself -------------------------------------------------------------------
template <int I, int N, class F>
P2_HD void phi_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    phi_static_for<I + 1, N>(f);
  }
}

// in-register 2^LOGR-point DFT on phi_t registers whose components are bounded by IN_MA / IN_MB;
// DIT: bit-reversed in -> natural out; DIF: natural in -> bit-reversed out (same index conventions as
// ntt.hip dft_regs).  All internal twiddles are powers of two; normalisations are placed at compile time.
template <int LOGR, int DIT, bool INV, uint64_t IN_MA, uint64_t IN_MB>
struct PhiNet {
  static constexpr PhiSched sched = phi_sched<LOGR, DIT, INV>(PhiBound{IN_MA, IN_MB});
  static_assert(sched.ok, "butterfly network does not fit int64 components");
  static P2_HD void run(phi_t (&v)[1 << LOGR]) { run_part<0, LOGR, -1>(v); }
  // layers [LC0, LC1) in execution order, restricted (HALF = 0 / 1) to the butterflies whose registers
  // all lie in the lower / upper half of the register array (HALF = -1: no restriction).  The network is
  // the same dataflow however it is cut: a DIT network's layers below the top one never cross the two
  // halves, so a caller can finish one half before it even loads the other (register pressure).
  template <int LC0, int LC1, int HALF>
  static P2_HD void run_part(phi_t (&v)[1 << LOGR]) {
    constexpr int R = 1 << LOGR;
    phi_static_for<LC0, LC1>([&](auto lc) {
      constexpr int lam = DIT ? decltype(lc)::value : LOGR - 1 - decltype(lc)::value;
      static_assert(HALF < 0 || lam < LOGR - 1, "the top layer crosses the halves");
      phi_static_for<0, R>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr ((j & (1 << lam)) == 0 && (HALF < 0 || (j >> (LOGR - 1)) == HALF)) {
          constexpr int k = j | (1 << lam);
          constexpr int e = phi_tw_exp(lam, j & ((1 << lam) - 1), INV);
          if constexpr (sched.pre[lam][j]) v[j] = phi_norm(v[j]);
          if constexpr (sched.pre[lam][k]) v[k] = phi_norm(v[k]);
          if constexpr (DIT) {
            const phi_t u = v[j], t = phi_mul_pow2<e>(v[k]);
            v[j] = phi_add(u, t);
            v[k] = phi_sub(u, t);
          } else {
            const phi_t u = v[j], x = v[k];
            v[j] = phi_add(u, x);
            phi_t dlt = phi_sub(u, x);
            if constexpr (sched.mid[lam][k]) dlt = phi_norm(dlt);
            v[k] = phi_mul_pow2<e>(dlt);
          }
        }
      });
    });
  }
  // butterfly (J, J + R/2) of the TOP layer of a DIT network (the last one executed)
  template <int J>
  static P2_HD void dit_top(phi_t (&v)[1 << LOGR]) {
    static_assert(DIT == 1, "DIT networks only");
    constexpr int lam = LOGR - 1, k = J | (1 << lam);
    constexpr int e = phi_tw_exp(lam, J & ((1 << lam) - 1), INV);
    if constexpr (sched.pre[lam][J]) v[J] = phi_norm(v[J]);
    if constexpr (sched.pre[lam][k]) v[k] = phi_norm(v[k]);
    const phi_t u = v[J], t = phi_mul_pow2<e>(v[k]);
    v[J] = phi_add(u, t);
    v[k] = phi_sub(u, t);
  }
};

}  // namespace p2
