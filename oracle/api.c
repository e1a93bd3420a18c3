/* oracle/api.c -- thin stage-level wrappers for the parity tests.  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include "circuit.h"
#include <stdlib.h>
#include <string.h>

void orc_ntt(uint64_t *a, unsigned lg, int inverse) {
  if (inverse) intt(a, lg);
  else ntt(a, lg);
}
/* PolynomialCoeffs::lde(rate_bits).coset_fft(g): natural-order LDE values */
void orc_coset_lde(const uint64_t *coeffs, unsigned d, unsigned rate_bits, uint64_t *out) {
  size_t n = (size_t)1 << d, N = n << rate_bits;
  memcpy(out, coeffs, 8 * n);
  memset(out + n, 0, 8 * (N - n));
  coset_ntt(out, d + rate_bits, GL_GENERATOR);
}
void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]) { keccak256(in, len, out); }
void orc_keccak_permutation(uint64_t st[12]) { keccak_permutation(st); }
void orc_poseidon_permute(uint64_t st[12]) { poseidon_permute(st); }
void orc_poseidon_round_constants(uint64_t out[360]) { memcpy(out, poseidon_round_constants(), 360 * 8); }
void orc_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]) { poseidon_hash_no_pad(in, n, out); }
void orc_set_hasher(int hasher) { g_hasher = hasher == 1; } /* for the stage-level helpers below */
void orc_commit_values(const uint64_t *vals, size_t ncols, unsigned d, unsigned rate_bits, unsigned cap_h, uint8_t *cap) {
  batch_t b;
  batch_from_values(&b, vals, ncols, d, rate_bits, cap_h);
  for (size_t i = 0; i < ((size_t)1 << cap_h); i++) memcpy(cap + DIGEST_BYTES * i, b.tree.cap[i].b, DIGEST_BYTES);
  batch_free(&b);
}
void orc_merkle_cap(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_h, uint8_t *cap) {
  merkle_t t;
  merkle_build(&t, leaves, n_leaves, leaf_len, cap_h);
  for (size_t i = 0; i < ((size_t)1 << cap_h); i++) memcpy(cap + DIGEST_BYTES * i, t.cap[i].b, DIGEST_BYTES);
  merkle_free(&t);
}
int orc_gate_eval(uint32_t kind, const uint32_t params[4], const uint64_t *wires, const uint64_t *consts,
                  const uint64_t pi_hash[4], uint64_t *out) {
  gate_t g;
  memset(&g, 0, sizeof g);
  g.kind = kind;
  memcpy(g.p, params, 16);
  g.num_constraints = gate_num_constraints(kind, params);
  gate_eval_base(&g, wires, consts, pi_hash, out);
  return (int)g.num_constraints;
}
int orc_gate_eval_ext(uint32_t kind, const uint32_t params[4], const uint64_t *wires, const uint64_t *consts,
                      const uint64_t pi_hash[4], uint64_t *out) {
  gate_t g;
  memset(&g, 0, sizeof g);
  g.kind = kind;
  memcpy(g.p, params, 16);
  g.num_constraints = gate_num_constraints(kind, params);
  ext_t pih[4];
  for (int i = 0; i < 4; i++) pih[i] = ext_from(pi_hash[i]);
  gate_eval_ext(&g, (const ext_t *)wires, (const ext_t *)consts, pih, (ext_t *)out);
  return (int)g.num_constraints;
}
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(gl_canon(a), gl_canon(b)); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
void orc_challenger_squeeze(const uint64_t *obs, size_t n, uint64_t *out, size_t m) {
  challenger_t c;
  ch_init(&c);
  ch_observe_many(&c, obs, n);
  for (size_t i = 0; i < m; i++) out[i] = ch_get(&c);
}
