// hostproof.hpp -- host-side pieces shared by the verifier (verify.hip) and the proof
// (de)compression (proofio.hip): a bounds-checked byte cursor, KeccakHash<25> leaf / path hashing,
// alpha-power reduction and the closed-form coset interpolation of a FRI fold.
#pragma once
#include "circuit.hpp"

namespace p2 {

// bounds-checked cursor over the proof bytes
struct Cursor {
  const uint8_t *p;
  size_t len, at = 0;
  bool ok = true;
  const uint8_t *take(size_t n) {
    if (!ok || n > len - at) {
      ok = false;
      return nullptr;
    }
    const uint8_t *q = p + at;
    at += n;
    return q;
  }
  uint64_t u64() {
    uint64_t v = 0;
    if (const uint8_t *q = take(8)) memcpy(&v, q, 8);
    return v;
  }
  gl_t felt() {  // canonical field element
    uint64_t v = u64();
    if (v >= GL_P) ok = false;
    return v;
  }
  ext_t ext() {
    gl_t a = felt(), b = felt();
    return ext_make(a, b);
  }
  dig_t digest() {
    dig_t d;
    memset(&d, 0, sizeof d);
    if (const uint8_t *q = take(hh_bytes())) memcpy(d.w, q, hh_bytes());
    // PoseidonHash digests are 4 field elements: w and w + p hash alike, so only the canonical encoding is a proof
    if (g_hh.kind)
      for (int i = 0; i < 4; i++)
        if (d.w[i] >= GL_P) ok = false;
    return d;
  }
  void digests(std::vector<dig_t> &v, size_t n) {
    v.resize(n);
    for (auto &d : v) d = digest();
  }
};

// hash/merkle_proofs.rs verify_merkle_proof_to_cap
inline bool merkle_path_ok(const gl_t *leaf, size_t n, size_t index, const std::vector<dig_t> &siblings,
                    const std::vector<dig_t> &cap) {
  dig_t cur = leaf_digest(leaf, n);
  for (const dig_t &s : siblings) {
    cur = (index & 1) ? node_digest(s, cur) : node_digest(cur, s);
    index >>= 1;
  }
  return index < cap.size() && dig_eq(cur, cap[index]);
}

// sum_j alpha^j v_j
template <class It>
ext_t reduce_with_powers(It first, It last, ext_t alpha) {
  ext_t acc = ext_from(0);
  while (last != first) {
    --last;
    acc = ext_add(ext_mul(acc, alpha), *last);
  }
  return acc;
}

// Value at beta of the degree < a interpolant through (s g^j, y_j), j < a = 2^ab, g of order a.
// The nodes are the roots of X^a - s^a, so the barycentric weights are x_j / (a s^a):
//   P(beta) = (beta^a - s^a) / (a s^a) * sum_j y_j x_j / (beta - x_j).
// (fri/verifier.rs compute_evaluation does the same interpolation with a generic routine.)
inline ext_t interpolate_coset(gl_t s, unsigned ab, const ext_t *y_natural, ext_t beta) {
  const uint32_t a = 1u << ab;
  const gl_t g = gl_root(ab);
  gl_t sa = s;
  ext_t ba = beta;
  for (unsigned i = 0; i < ab; i++) {
    sa = gl_sqr(sa);
    ba = ext_mul(ba, ba);
  }
  ext_t acc = ext_from(0);
  gl_t x = s;
  for (uint32_t j = 0; j < a; j++) {
    const ext_t diff = ext_sub(beta, ext_from(x));
    if (diff.c0 == 0 && diff.c1 == 0) return y_natural[j];
    acc = ext_add(acc, ext_mul(ext_scale(y_natural[j], x), ext_inv(diff)));
    x = gl_mul(x, g);
  }
  const gl_t scale = gl_inv(gl_mul((gl_t)a, sa));
  return ext_mul(ext_scale(ext_sub(ba, ext_from(sa)), scale), acc);
}

}  // namespace p2
