// synth.cpp -- synthetic circuits with the reference's shape (workload generator
// for tests and bench.py; host-only C++, built as libp2synth.so).
//
// The named ACIR programs of BASELINE.json cannot be compiled here (no nargo /
// Rust), so the benchmark inputs are circuits with the shape the reference's
// translator fixes -- CircuitConfig::wide_ecc_config():
// plonky2-backend/src/circuit_translation/mod.rs:69 (234 wires, 80 routed,
// 2 challenges, rate 8, cap 2^4, 28 queries, 16 PoW bits, arity 16) -- and the
// gate kinds it emits (SURVEY.md Appendix A + the five custom gates under
// plonky2-backend/src/plonky2_ecdsa/biguint/gates/).  This file plays the part
// of `CircuitBuilder::build()` + `generate_partial_witness` for those circuits:
// gate sorting and selector groups (gates/selectors.rs), copy-constraint sigma
// polynomials (plonk/permutation_argument.rs), and satisfying wire values
// produced with the same formulas as the gates' generators, e.g.
// arithmetic_u32.rs:376-426, add_many_u32.rs:329-378, subtraction_u32.rs:298-343,
// range_check_u32.rs:198-220, comparison.rs:439-537.
#include "gl.hpp"
#include "poseidon.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

using namespace p2;

namespace {

enum {
  G_NOOP = 0, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC, G_BASE_SUM, G_RANDOM_ACCESS, G_POSEIDON,
  G_U32_ARITHMETIC, G_U32_ADD_MANY, G_U32_SUBTRACTION, G_U32_RANGE_CHECK, G_COMPARISON
};

struct GateDef {
  uint32_t kind;
  uint32_t p[4];
  std::string id;
  uint32_t degree, ncons, nconst;
  double weight;  // share of rows
};

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  gl_t field() { return gl_canon(next()); }  // splitmix64 reduced mod p
  uint64_t u32() { return next() >> 32; }
  uint64_t below(uint64_t m) { return next() % m; }
};

GateDef mk(uint32_t kind, uint32_t p0, uint32_t p1, uint32_t p2, const std::string &id, uint32_t deg, uint32_t ncons,
           uint32_t nconst, double w) {
  GateDef g;
  g.kind = kind;
  g.p[0] = p0; g.p[1] = p1; g.p[2] = p2; g.p[3] = 0;
  g.id = id; g.degree = deg; g.ncons = ncons; g.nconst = nconst; g.weight = w;
  return g;
}

struct Cell { uint32_t row, col; };

struct Builder {
  unsigned d;
  size_t n;
  uint32_t W, R;
  Rng rng;
  std::vector<gl_t> wires;      // [W][n]
  std::vector<uint32_t> parent; // union-find over R*n routed cells, index col*n+row
  std::vector<Cell> pool_u32, pool_f;

  gl_t &w(uint32_t row, uint32_t col) { return wires[(size_t)col * n + row]; }
  uint32_t find(uint32_t x) {
    while (parent[x] != x) {
      parent[x] = parent[parent[x]];
      x = parent[x];
    }
    return x;
  }
  void connect(Cell a, Cell b) {
    uint32_t x = find(a.col * (uint32_t)n + a.row), y = find(b.col * (uint32_t)n + b.row);
    if (x != y) parent[x] = y;
  }
  void pool_add(std::vector<Cell> &pool, Cell c) {
    if (c.col >= R) return;
    if (pool.size() < (1u << 16)) pool.push_back(c);
    else pool[rng.below(pool.size())] = c;
  }
  // obtain an input value: copy-constrained to an earlier cell half of the time
  gl_t take(std::vector<Cell> &pool, uint32_t row, uint32_t col, bool u32) {
    gl_t v;
    if (col < R && !pool.empty() && (rng.next() & 1)) {
      Cell src = pool[rng.below(pool.size())];
      v = w(src.row, src.col);
      connect(Cell{row, col}, src);
    } else {
      v = u32 ? rng.u32() : rng.field();
    }
    w(row, col) = v;
    return v;
  }
  gl_t take_u32(uint32_t row, uint32_t col) { return take(pool_u32, row, col, true); }
  gl_t take_f(uint32_t row, uint32_t col) { return take(pool_f, row, col, false); }
  void out_u32(uint32_t row, uint32_t col, gl_t v) {
    w(row, col) = v;
    pool_add(pool_u32, Cell{row, col});
  }
  void out_f(uint32_t row, uint32_t col, gl_t v) {
    w(row, col) = v;
    pool_add(pool_f, Cell{row, col});
  }
};

void fill_row(Builder &b, const GateDef &g, uint32_t row, gl_t *lc /* local constants, nconst */) {
  Rng &rng = b.rng;
  switch (g.kind) {
  case G_NOOP:
    break;
  case G_CONSTANT:
    for (uint32_t i = 0; i < g.p[0]; i++) {
      lc[i] = (row == 1) ? i : rng.field();
      b.out_f(row, i, lc[i]);
      if (lc[i] < (1ull << 32)) b.pool_add(b.pool_u32, Cell{row, i});
    }
    break;
  case G_PUBLIC_INPUT:
    for (uint32_t i = 0; i < 4; i++) b.w(row, i) = 0;  // hash of no public inputs
    break;
  case G_ARITHMETIC: {
    lc[0] = (rng.next() & 3) ? 1 : rng.field();
    lc[1] = (rng.next() & 3) ? 1 : rng.field();
    for (uint32_t i = 0; i < g.p[0]; i++) {
      gl_t m0 = b.take_f(row, 4 * i), m1 = b.take_f(row, 4 * i + 1), ad = b.take_f(row, 4 * i + 2);
      gl_t o = gl_add(gl_mul(gl_mul(m0, m1), lc[0]), gl_mul(ad, lc[1]));
      b.out_f(row, 4 * i + 3, o);
    }
    break;
  }
  case G_BASE_SUM: {
    uint32_t B = g.p[0], L = g.p[1];
    uint64_t v = b.take_u32(row, 0);
    for (uint32_t i = 0; i < L; i++) {
      b.w(row, 1 + i) = v % B;
      v /= B;
    }
    break;
  }
  case G_RANDOM_ACCESS: {
    uint32_t bits = g.p[0], copies = g.p[1], extra = g.p[2], vec = 1u << bits;
    uint32_t routed = (2 + vec) * copies + extra;
    for (uint32_t c = 0; c < copies; c++) {
      uint32_t base = (2 + vec) * c;
      uint32_t idx = (uint32_t)rng.below(vec);
      b.w(row, base) = idx;
      gl_t claimed = 0;
      for (uint32_t i = 0; i < vec; i++) {
        gl_t it = b.take_f(row, base + 2 + i);
        if (i == idx) claimed = it;
      }
      b.out_f(row, base + 1, claimed);
      for (uint32_t k = 0; k < bits; k++) b.w(row, routed + c * bits + k) = (idx >> k) & 1;
    }
    for (uint32_t i = 0; i < extra; i++) {
      lc[i] = rng.field();
      b.out_f(row, (2 + vec) * copies + i, lc[i]);
    }
    break;
  }
  case G_U32_ARITHMETIC: {  // arithmetic_u32.rs:376-426
    uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      uint64_t m0 = b.take_u32(row, 6 * i), m1 = b.take_u32(row, 6 * i + 1), ad = b.take_u32(row, 6 * i + 2);
      uint64_t o = gl_add(gl_mul(m0, m1), ad);
      uint64_t hi = o >> 32, lo = o & 0xFFFFFFFFull;
      b.out_u32(row, 6 * i + 3, lo);
      b.out_u32(row, 6 * i + 4, hi);
      uint64_t diff = 0xFFFFFFFFull - hi;
      b.w(row, 6 * i + 5) = diff ? gl_inv(diff) : 0;
      for (uint32_t j = 0; j < 32; j++) {
        b.w(row, 6 * ops + 32 * i + j) = o & 3;
        o >>= 2;
      }
    }
    break;
  }
  case G_U32_ADD_MANY: {  // add_many_u32.rs:329-378
    uint32_t na = g.p[0], ops = g.p[1];
    for (uint32_t i = 0; i < ops; i++) {
      uint32_t base = (na + 3) * i;
      gl_t sum = 0;
      for (uint32_t j = 0; j < na; j++) sum = gl_add(sum, b.take_u32(row, base + j));
      sum = gl_add(sum, b.take_u32(row, base + na));
      uint64_t res = sum & 0xFFFFFFFFull, carry = sum >> 32;
      b.out_u32(row, base + na + 1, res);
      b.out_u32(row, base + na + 2, carry);
      for (uint32_t j = 0; j < 16; j++) b.w(row, (na + 3) * ops + 18 * i + j) = (res >> (2 * j)) & 3;
      for (uint32_t j = 0; j < 2; j++) b.w(row, (na + 3) * ops + 18 * i + 16 + j) = (carry >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_SUBTRACTION: {  // subtraction_u32.rs:298-343
    uint32_t ops = g.p[0];
    for (uint32_t i = 0; i < ops; i++) {
      gl_t x = b.take_u32(row, 5 * i), y = b.take_u32(row, 5 * i + 1);
      gl_t bin = rng.next() & 1;
      b.w(row, 5 * i + 2) = bin;
      gl_t init = gl_sub(gl_sub(x, y), bin);
      gl_t bout = init > (1ull << 32) ? 1 : 0;
      gl_t res = gl_add(init, gl_mul(bout, 1ull << 32));
      b.out_u32(row, 5 * i + 3, res);
      b.w(row, 5 * i + 4) = bout;
      for (uint32_t j = 0; j < 16; j++) b.w(row, 5 * ops + 16 * i + j) = (res >> (2 * j)) & 3;
    }
    break;
  }
  case G_U32_RANGE_CHECK: {  // range_check_u32.rs:198-220
    uint32_t nl = g.p[0];
    for (uint32_t i = 0; i < nl; i++) {
      uint64_t v = b.take_u32(row, i);
      for (uint32_t j = 0; j < 16; j++) b.w(row, nl + 16 * i + j) = (v >> (2 * j)) & 3;
    }
    break;
  }
  case G_COMPARISON: {  // comparison.rs:439-537
    uint32_t nb = g.p[0], nc = g.p[1], cb = (nb + nc - 1) / nc;
    uint64_t a = b.take_u32(row, 0), c2 = (rng.next() & 7) ? b.take_u32(row, 1) : (b.w(row, 1) = a);
    if (nb < 32) { a &= (1ull << nb) - 1; c2 &= (1ull << nb) - 1; b.w(row, 0) = a; b.w(row, 1) = c2; }
    b.w(row, 2) = a <= c2 ? 1 : 0;
    uint64_t cs = 1ull << cb;
    gl_t msd = 0;
    uint64_t ta = a, tb = c2;
    for (uint32_t i = 0; i < nc; i++) {
      gl_t f = ta % cs, s = tb % cs;
      ta /= cs;
      tb /= cs;
      b.w(row, 4 + i) = f;
      b.w(row, 4 + nc + i) = s;
      b.w(row, 4 + 2 * nc + i) = (f == s) ? 1 : gl_inv(gl_sub(s, f));
      b.w(row, 4 + 3 * nc + i) = (f == s) ? 1 : 0;
      if (f != s) {
        msd = gl_sub(s, f);
        b.w(row, 4 + 4 * nc + i) = 0;
      } else {
        b.w(row, 4 + 4 * nc + i) = msd;
      }
    }
    b.w(row, 3) = msd;
    uint64_t v = gl_add(cs, msd);
    for (uint32_t i = 0; i < cb + 1; i++) {
      b.w(row, 4 + 5 * nc + i) = v & 1;
      v >>= 1;
    }
    break;
  }
  }
}

}  // namespace

extern "C" {

// The 135 wires of ONE PoseidonGate row (plonky2 gates/poseidon.rs layout: inputs 0..11, outputs 12..23, swap 24, deltas 25..28,
// S-box inputs of full rounds 1..3 at 29..64, of the 22 partial rounds at 65..86, of full rounds 26..29 at 87..134) for the given
// twelve inputs with swap = 0: what the gate's generator derives.  Used by translate.py's build() for the in-circuit hash
// of the public inputs (circuit_builder.rs build(): hash_n_to_hash_no_pad + PublicInputGate).
void p2synth_poseidon_gate_row(const uint64_t *in12, uint64_t *wires135) {
  gl_t prc[360];
  poseidon_round_constants_host(prc);
  gl_t st[12];
  for (int i = 0; i < 135; i++) wires135[i] = 0;
  for (int i = 0; i < 12; i++) wires135[i] = st[i] = in12[i];
  for (int rd = 0; rd < 30; rd++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], prc[12 * rd + i]);
    if (rd < 4 || rd >= 26) {
      if (rd != 0) {
        const int base = rd < 4 ? 29 + 12 * (rd - 1) : 87 + 12 * (rd - 26);
        for (int i = 0; i < 12; i++) wires135[base + i] = st[i];
      }
      for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(st[i]);
    } else {
      wires135[65 + (rd - 4)] = st[0];
      st[0] = poseidon_sbox(st[0]);
    }
    poseidon_mds(st);
  }
  for (int i = 0; i < 12; i++) wires135[12 + i] = st[i];
}


// mix: "arith" | "sha" | "ecdsa" | "grammar".  Returns 0 ok.  Outputs are malloc'ed; free
// with p2synth_free.  wires_out is [num_wires][2^d] column-major.
// num_pi > 0 adds what `build()` adds for public inputs: PoseidonGate rows hashing them (overwrite-mode
// sponge, 8 per permutation) and the PublicInputGate row wired to the hash; pis_out receives the values.
// num_wires: 234 = CircuitConfig::wide_ecc_config() (the translator's shape, mod.rs:69), 135 =
// standard_recursion_config() (used by the reference's memory tests, test_memory_operations.rs:160,389);
// the custom gates size themselves from it exactly as their `num_ops(config)` do.
// flags bit 0: randomise only the ROUTED unused wires of the PublicInputGate row (the rest stay zero), so that the
// row-local generators (N1) reproduce the whole matrix from the routed columns; default (0) = what plonky2's
// build() does: every wire of that row after the public-inputs hash gets a random value.
int p2synth_make2(unsigned d, const char *mix, uint64_t seed, uint32_t num_pi, uint32_t num_wires, uint32_t flags,
                  uint8_t **blob_out, size_t *blob_len, uint64_t **wires_out, uint32_t *num_wires_out, uint64_t *pis_out);
int p2synth_make(unsigned d, const char *mix, uint64_t seed, uint32_t num_pi, uint32_t num_wires, uint8_t **blob_out,
                 size_t *blob_len, uint64_t **wires_out, uint32_t *num_wires_out, uint64_t *pis_out) {
  return p2synth_make2(d, mix, seed, num_pi, num_wires, 0, blob_out, blob_len, wires_out, num_wires_out, pis_out);
}
int p2synth_make2(unsigned d, const char *mix, uint64_t seed, uint32_t num_pi, uint32_t num_wires, uint32_t flags,
                  uint8_t **blob_out, size_t *blob_len, uint64_t **wires_out, uint32_t *num_wires_out, uint64_t *pis_out) {
  const uint32_t W = num_wires ? num_wires : 234;
  const uint32_t R = 80, K = 2, QF = 8, RATE_BITS = 3, CAP_H = 4, POW_BITS = 16, QUERIES = 28;
  if (W != 234 && W != 135) return -5;
  if (d < 5 || d > 24) return -1;
  size_t n = (size_t)1 << d;
  std::string m(mix ? mix : "arith");
  const std::string ph = ", _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }";
  std::vector<GateDef> gates;
  gates.push_back(mk(G_NOOP, 0, 0, 0, "NoopGate", 0, 0, 0, 0));
  gates.push_back(mk(G_CONSTANT, 2, 0, 0, "ConstantGate { num_consts: 2 }", 1, 2, 2, 0));
  gates.push_back(mk(G_PUBLIC_INPUT, 0, 0, 0, "PublicInputGate", 1, 4, 0, 0));
  if (m == "arith") {
    gates.push_back(mk(G_ARITHMETIC, 20, 0, 0, "ArithmeticGate { num_ops: 20 }", 3, 20, 2, 1.0));
  } else if (m == "sha") {
    gates.push_back(mk(G_ARITHMETIC, 20, 0, 0, "ArithmeticGate { num_ops: 20 }", 3, 20, 2, 0.75));
    gates.push_back(mk(G_BASE_SUM, 2, 32, 0, "BaseSumGate { num_limbs: 32 } + Base: 2", 2, 33, 0, 0.25));
  } else if (m == "ecdsa") {
    gates.push_back(mk(G_ARITHMETIC, 20, 0, 0, "ArithmeticGate { num_ops: 20 }", 3, 20, 2, 0.25));
    gates.push_back(mk(G_BASE_SUM, 2, 32, 0, "BaseSumGate { num_limbs: 32 } + Base: 2", 2, 33, 0, 0.05));
    gates.push_back(mk(G_BASE_SUM, 4, 16, 0, "BaseSumGate { num_limbs: 16 } + Base: 4", 4, 17, 0, 0.10));
    gates.push_back(mk(G_RANDOM_ACCESS, 4, 4, 2,
                       "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2" + ph + "<D=2>", 5, 26, 2, 0.05));
    // arithmetic_u32.rs:39-42, add_many_u32.rs:43-48, subtraction_u32.rs:39-43: ops = min(wires / per-op wires, routed / per-op routed)
    const uint32_t ua = std::min(W / 38, R / 6), am = std::min(W / 24, R / 6), us = std::min(W / 21, R / 5);
    const uint32_t rcl = W >= 136 ? 8 : 7;  // U32RangeCheckGate{8} needs 136 wires
    auto S = [](uint32_t v) { return std::to_string(v); };
    gates.push_back(mk(G_U32_ARITHMETIC, ua, 0, 0, "U32ArithmeticGate { num_ops: " + S(ua) + ph, 4, ua * 36, 0, 0.20));
    gates.push_back(mk(G_U32_ADD_MANY, 3, am, 0, "U32AddManyGate { num_addends: 3, num_ops: " + S(am) + ph, 4, am * 21, 0, 0.10));
    gates.push_back(mk(G_U32_SUBTRACTION, us, 0, 0, "U32SubtractionGate { num_ops: " + S(us) + ph, 4, us * 19, 0, 0.10));
    gates.push_back(mk(G_U32_RANGE_CHECK, rcl, 0, 0, "U32RangeCheckGate { num_input_limbs: " + S(rcl) + ph, 4, rcl * 17, 0, 0.10));
    gates.push_back(mk(G_COMPARISON, 32, 16, 0, "ComparisonGate { num_bits: 32, num_chunks: 16" + ph + "<D=2>", 4, 88, 0, 0.05));
  } else if (m == "grammar") {
    // BASELINE configs[4] ("zk-grammar medium Noir project"; SURVEY 8(d): arith + sha mix incl. RandomAccess / memory
    // ops): field arithmetic, bit and base-4 decompositions, and memory reads -- the translator turns every ACIR
    // MemoryOp read into a RandomAccessGate lookup (circuit_translation/memory_translator.rs:118-122)
    gates.push_back(mk(G_ARITHMETIC, 20, 0, 0, "ArithmeticGate { num_ops: 20 }", 3, 20, 2, 0.55));
    gates.push_back(mk(G_BASE_SUM, 2, 32, 0, "BaseSumGate { num_limbs: 32 } + Base: 2", 2, 33, 0, 0.20));
    gates.push_back(mk(G_BASE_SUM, 4, 16, 0, "BaseSumGate { num_limbs: 16 } + Base: 4", 4, 17, 0, 0.05));
    gates.push_back(mk(G_RANDOM_ACCESS, 4, 4, 2,
                       "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2" + ph + "<D=2>", 5, 26, 2, 0.20));
  } else {
    return -2;
  }
  const uint32_t nperm = (num_pi + 7) / 8;
  if (num_pi > 64 || (size_t)nperm + 8 > n) return -4;
  if (num_pi)
    gates.push_back(mk(G_POSEIDON, 0, 0, 0,
                       "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", 7, 123, 0, 0));
  // circuit_builder.rs build(): gates sorted by (degree, id)
  std::sort(gates.begin(), gates.end(), [](const GateDef &a, const GateDef &b) {
    return a.degree != b.degree ? a.degree < b.degree : a.id < b.id;
  });
  uint32_t ng = (uint32_t)gates.size();
  // gates/selectors.rs selector_polynomials, max_degree = quotient_degree_factor + 1
  const uint32_t max_degree = QF + 1;
  std::vector<uint32_t> gstart(ng), gend(ng), gsel(ng);
  uint32_t num_selectors;
  if (gates.back().degree + ng - 1 <= max_degree) {
    num_selectors = 1;
    for (uint32_t i = 0; i < ng; i++) { gstart[i] = 0; gend[i] = ng; gsel[i] = 0; }
  } else {
    uint32_t start = 0, grp = 0;
    while (start < ng) {
      uint32_t size = 0;
      while (start + size < ng && size + gates[start + size].degree < max_degree) size++;
      for (uint32_t i = start; i < start + size; i++) { gstart[i] = start; gend[i] = start + size; gsel[i] = grp; }
      start += size;
      grp++;
    }
    num_selectors = grp;
  }
  uint32_t max_nconst = 0;
  for (auto &g : gates) max_nconst = std::max(max_nconst, g.nconst);
  const uint32_t NC = num_selectors + max_nconst;
  auto gate_index = [&](uint32_t kind, uint32_t p0) {
    for (uint32_t i = 0; i < ng; i++)
      if (gates[i].kind == kind && (kind != G_BASE_SUM || gates[i].p[0] == p0)) return i;
    return 0u;
  };

  Builder b;
  b.d = d; b.n = n; b.W = W; b.R = R;
  b.rng.s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567;
  b.wires.assign((size_t)W * n, 0);
  b.parent.resize((size_t)R * n);
  std::iota(b.parent.begin(), b.parent.end(), 0u);

  std::vector<gl_t> constants((size_t)NC * n, 0);
  // row -> gate
  std::vector<uint32_t> row_gate(n);
  size_t pad = std::max<size_t>(1, n / 64);
  {
    std::vector<uint32_t> cand;
    std::vector<double> cum;
    double acc = 0;
    for (uint32_t i = 0; i < ng; i++)
      if (gates[i].weight > 0) { cand.push_back(i); acc += gates[i].weight; cum.push_back(acc); }
    uint32_t gi_noop = gate_index(G_NOOP, 0);
    for (size_t r = 0; r < n; r++) {
      if (r == 0) row_gate[r] = gate_index(G_PUBLIC_INPUT, 0);
      else if (r == 1) row_gate[r] = gate_index(G_CONSTANT, 0);
      else if (r < 2 + nperm) row_gate[r] = gate_index(G_POSEIDON, 0);
      else if (r >= n - pad) row_gate[r] = gi_noop;
      else {
        double u = (double)(b.rng.next() >> 11) / 9007199254740992.0 * acc;
        size_t k = 0;
        while (k + 1 < cand.size() && u >= cum[k]) k++;
        row_gate[r] = cand[k];
      }
    }
  }
  // fill rows (row 1 first so the constants 0/1 exist before other rows connect to them)
  gl_t lc[8];
  gl_t pi_hash[4] = {0, 0, 0, 0};
  std::vector<gl_t> pis(num_pi);
  std::vector<size_t> order;
  order.push_back(1);
  for (size_t r = 2; r < 2 + (size_t)nperm; r++) order.push_back(r);
  order.push_back(0);
  for (size_t r = 2 + nperm; r < n; r++) order.push_back(r);
  gl_t prc[360];
  if (num_pi) poseidon_round_constants_host(prc);
  for (size_t r : order) {
    const GateDef &g = gates[row_gate[r]];
    memset(lc, 0, sizeof lc);
    if (g.kind == G_POSEIDON) {
      // in-circuit hash_n_to_hash_no_pad (hash/hashing.rs): PoseidonGate row = one permutation, swap = 0
      const uint32_t c = (uint32_t)r - 2, row = (uint32_t)r;
      gl_t st[12];
      for (uint32_t i = 0; i < 12; i++) {
        uint32_t k = 8 * c + i;
        if (i < 8 && k < num_pi) {
          pis[k] = b.rng.field();
          b.out_f(row, i, pis[k]);  // the public-input target; later gates may copy from it
          st[i] = pis[k];
        } else if (c == 0) {
          b.w(row, i) = 0;
          b.connect(Cell{row, i}, Cell{1, 0});
          st[i] = 0;
        } else {
          st[i] = b.w(row - 1, 12 + i);
          b.w(row, i) = st[i];
          b.connect(Cell{row, i}, Cell{row - 1, 12 + i});
        }
      }
      b.w(row, 24) = 0;  // swap
      b.connect(Cell{row, 24}, Cell{1, 0});
      for (uint32_t i = 0; i < 4; i++) b.w(row, 25 + i) = 0;  // deltas
      for (int rd = 0; rd < 30; rd++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], prc[12 * rd + i]);
        if (rd < 4 || rd >= 26) {
          if (rd != 0) {
            uint32_t base = rd < 4 ? 29 + 12 * (rd - 1) : 87 + 12 * (rd - 26);
            for (int i = 0; i < 12; i++) b.w(row, base + i) = st[i];
          }
          for (int i = 0; i < 12; i++) st[i] = poseidon_sbox(st[i]);
        } else {
          b.w(row, 65 + (rd - 4)) = st[0];
          st[0] = poseidon_sbox(st[0]);
        }
        poseidon_mds(st);
      }
      for (uint32_t i = 0; i < 12; i++) b.out_f(row, 12 + i, st[i]);
      if (c == nperm - 1)
        for (int i = 0; i < 4; i++) pi_hash[i] = st[i];
    } else if (g.kind == G_PUBLIC_INPUT) {
      for (uint32_t i = 0; i < 4; i++) {
        b.w((uint32_t)r, i) = pi_hash[i];
        if (num_pi) b.connect(Cell{(uint32_t)r, i}, Cell{1 + nperm, 12 + i});
        else b.connect(Cell{(uint32_t)r, i}, Cell{1, 0});  // hash of no public inputs = constant zero
      }
      // circuit_builder.rs randomize_unused_pi_wires: build() hangs a RandomValueGenerator on every
      // other wire of this row, routed or not (visible in the reference's own proofs,
      // tests/golden/reference_proofs.py: no wire column of a real witness is zero in every row, the unused
      // ones hold exactly this one value).  The non-routed ones come from a stream of their own, so the rest
      // of the matrix does not depend on flags bit 0 (routed wires only: what the row-local generators can
      // rebuild from the routed columns).
      for (uint32_t i = 4; i < R; i++) b.w((uint32_t)r, i) = b.rng.field();
      if (!(flags & 1)) {
        Rng extra;
        extra.s = seed * 0xD1B54A32D192ED03ULL + 0x7654321;
        for (uint32_t i = R; i < W; i++) b.w((uint32_t)r, i) = extra.field();
      }
    } else {
      fill_row(b, g, (uint32_t)r, lc);
    }
    uint32_t gi = row_gate[r];
    for (uint32_t s = 0; s < num_selectors; s++)
      constants[(size_t)s * n + r] = (num_selectors == 1 || s == gsel[gi]) ? gi : 0xFFFFFFFFull;
    for (uint32_t k = 0; k < g.nconst; k++) constants[(size_t)(num_selectors + k) * n + r] = lc[k];
  }
  if (pis_out)
    for (uint32_t i = 0; i < num_pi; i++) pis_out[i] = pis[i];

  // plonk/permutation_argument.rs: sigma maps each routed cell to the next cell of its partition
  // class; WirePartition lists a class row by row (wire_partition(): `for row.. for column..`), which
  // is the cycle order the reference's own proofs show (tests/golden/reference_proofs.py)
  std::vector<gl_t> k_is(R);
  k_is[0] = 1;
  for (uint32_t j = 1; j < R; j++) k_is[j] = gl_mul(k_is[j - 1], GL_GEN);
  std::vector<gl_t> sub(n);
  {
    gl_t wn = gl_root(d);
    sub[0] = 1;
    for (size_t i = 1; i < n; i++) sub[i] = gl_mul(sub[i - 1], wn);
  }
  std::vector<gl_t> sigmas((size_t)R * n);
  {
    size_t tot = (size_t)R * n;
    std::vector<uint32_t> root(tot), next(tot), last(tot, UINT32_MAX), first(tot, UINT32_MAX);
    for (size_t x = 0; x < tot; x++) root[x] = b.find((uint32_t)x);
    for (size_t row = 0; row < n; row++)
      for (uint32_t col = 0; col < R; col++) {
        const size_t x = (size_t)col * n + row;
        uint32_t rt = root[x];
        if (first[rt] == UINT32_MAX) first[rt] = (uint32_t)x;
        else next[last[rt]] = (uint32_t)x;
        last[rt] = (uint32_t)x;
      }
    for (size_t x = 0; x < tot; x++)
      if (last[root[x]] == x) next[x] = first[root[x]];
    for (size_t x = 0; x < tot; x++) {
      uint32_t y = next[x];
      uint32_t col = y / (uint32_t)n, row = y % (uint32_t)n;
      sigmas[x] = gl_mul(k_is[col], sub[row]);
    }
  }

  // FRI reduction strategy ConstantArityBits(4, 5) (fri/reduction_strategies.rs)
  std::vector<uint32_t> arity;
  {
    unsigned db = d;
    while (db > 5 && db + RATE_BITS - 4 >= CAP_H) { arity.push_back(4); db -= 4; }
  }

  // ---- blob ----
  size_t blen = 4 * 64 + (size_t)ng * 48 + 8 * ((size_t)R + (size_t)NC * n + (size_t)R * n);
  uint8_t *blob = (uint8_t *)malloc(blen);
  memset(blob, 0, 4 * 64);
  uint32_t h[64] = {0};
  h[0] = 0x43473250u; h[1] = 1; h[2] = d; h[3] = W; h[4] = R; h[5] = NC; h[6] = num_selectors; h[7] = K; h[8] = QF;
  h[9] = RATE_BITS; h[10] = CAP_H; h[11] = POW_BITS; h[12] = QUERIES; h[13] = (uint32_t)arity.size();
  for (size_t i = 0; i < arity.size(); i++) h[14 + i] = arity[i];
  h[22] = 0; h[23] = ng; h[24] = num_pi; h[25] = 0; h[26] = (R + QF - 1) / QF - 1;
  memcpy(blob, h, sizeof h);
  size_t off = sizeof h;
  for (uint32_t i = 0; i < ng; i++) {
    uint32_t gw[12] = {gates[i].kind, gates[i].p[0], gates[i].p[1], gates[i].p[2], gates[i].p[3], gsel[i], gstart[i],
                       gend[i], gates[i].ncons, gates[i].degree, gates[i].nconst, 0};
    memcpy(blob + off, gw, sizeof gw);
    off += sizeof gw;
  }
  memcpy(blob + off, k_is.data(), 8 * (size_t)R);
  off += 8 * (size_t)R;
  memcpy(blob + off, constants.data(), 8 * (size_t)NC * n);
  off += 8 * (size_t)NC * n;
  memcpy(blob + off, sigmas.data(), 8 * (size_t)R * n);
  off += 8 * (size_t)R * n;
  uint64_t *wo = (uint64_t *)malloc(8 * (size_t)W * n);
  memcpy(wo, b.wires.data(), 8 * (size_t)W * n);
  *blob_out = blob;
  *blob_len = blen;
  *wires_out = wo;
  if (num_wires_out) *num_wires_out = W;
  return off == blen ? 0 : -3;
}

void p2synth_free(void *p) { free(p); }
}
