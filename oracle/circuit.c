/* oracle/circuit.c -- see circuit.h.  TEST INFRASTRUCTURE ONLY. */
#include "circuit.h"
#include <stdlib.h>
#include <string.h>

/* ---- base-field instantiation ---- */
#define T gl_t
#define ADD gl_add
#define SUB gl_sub
#define MUL gl_mul
#define K(x) ((gl_t)(x))
#define NAME(x) x##_b
#include "gates_impl.inc"
#undef T
#undef ADD
#undef SUB
#undef MUL
#undef K
#undef NAME
/* ---- extension-field instantiation ---- */
#define T ext_t
#define ADD ext_add
#define SUB ext_sub
#define MUL ext_mul
#define K(x) ext_from((gl_t)(x))
#define NAME(x) x##_e
#include "gates_impl.inc"
#undef T
#undef ADD
#undef SUB
#undef MUL
#undef K
#undef NAME

void gate_eval_base(const gate_t *g, const gl_t *w, const gl_t *lc, const gl_t *pih, gl_t *out) {
  gate_eval_impl_b(g, w, lc, pih, out);
}
void gate_eval_ext(const gate_t *g, const ext_t *w, const ext_t *lc, const ext_t *pih, ext_t *out) {
  gate_eval_impl_e(g, w, lc, pih, out);
}
gl_t gate_filter_base(const circuit_t *c, uint32_t gi, gl_t s) { return gate_filter_impl_b(c, gi, s); }
ext_t gate_filter_ext(const circuit_t *c, uint32_t gi, ext_t s) { return gate_filter_impl_e(c, gi, s); }
void eval_gate_constraints_base(const circuit_t *c, const gl_t *crow, const gl_t *wires, const gl_t *pih, gl_t *out,
                                gl_t *scratch) {
  eval_gate_constraints_impl_b(c, crow, wires, pih, out, scratch);
}
void eval_gate_constraints_ext(const circuit_t *c, const ext_t *crow, const ext_t *wires, const ext_t *pih, ext_t *out,
                               ext_t *scratch) {
  eval_gate_constraints_impl_e(c, crow, wires, pih, out, scratch);
}

/* ---- gate metadata (num_constraints / degree / num_constants / num_wires) ---- */
uint32_t gate_num_constraints(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_NOOP: return 0;
  case G_CONSTANT: return p[0];
  case G_PUBLIC_INPUT: return 4;
  case G_ARITHMETIC: return p[0];
  case G_BASE_SUM: return 1 + p[1];
  case G_RANDOM_ACCESS: return p[1] * (p[0] + 2) + p[2];
  case G_POSEIDON: return 123;
  case G_U32_ARITHMETIC: return p[0] * (4 + 32);          /* arithmetic_u32.rs:281-283 */
  case G_U32_ADD_MANY: return p[1] * (3 + 18);            /* add_many_u32.rs:282-284 */
  case G_U32_SUBTRACTION: return p[0] * (3 + 16);         /* subtraction_u32.rs:226-228 */
  case G_U32_RANGE_CHECK: return p[0] * 17;               /* range_check_u32.rs:173-175 */
  case G_COMPARISON: return 6 + 5 * p[1] + (p[0] + p[1] - 1) / p[1]; /* comparison.rs:329-331 */
  }
  return 0;
}
uint32_t gate_degree(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_NOOP: return 0;
  case G_CONSTANT: return 1;
  case G_PUBLIC_INPUT: return 1;
  case G_ARITHMETIC: return 3;
  case G_BASE_SUM: return p[0];
  case G_RANDOM_ACCESS: return p[0] + 1;
  case G_POSEIDON: return 7;
  case G_U32_ARITHMETIC: return 4;
  case G_U32_ADD_MANY: return 4;
  case G_U32_SUBTRACTION: return 4;
  case G_U32_RANGE_CHECK: return 4;
  case G_COMPARISON: return 1u << ((p[0] + p[1] - 1) / p[1]); /* comparison.rs:325-327 */
  }
  return 0;
}
uint32_t gate_num_constants(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_CONSTANT: return p[0];
  case G_ARITHMETIC: return 2;
  case G_RANDOM_ACCESS: return p[2];
  }
  return 0;
}
uint32_t gate_num_wires(uint32_t kind, const uint32_t p[4]) {
  switch (kind) {
  case G_NOOP: return 0;
  case G_CONSTANT: return p[0];
  case G_PUBLIC_INPUT: return 4;
  case G_ARITHMETIC: return 4 * p[0];
  case G_BASE_SUM: return 1 + p[1];
  case G_RANDOM_ACCESS: return (2 + (1u << p[0])) * p[1] + p[2] + p[0] * p[1];
  case G_POSEIDON: return 135;
  case G_U32_ARITHMETIC: return p[0] * 38;
  case G_U32_ADD_MANY: return p[1] * (p[0] + 3 + 18);
  case G_U32_SUBTRACTION: return p[0] * 21;
  case G_U32_RANGE_CHECK: return p[0] * 17;
  case G_COMPARISON: return 4 + 5 * p[1] + (p[0] + p[1] - 1) / p[1] + 1;
  }
  return 0;
}

/* ---- blob ---- */
int circuit_parse(circuit_t *c, const uint8_t *blob, size_t len) {
  memset(c, 0, sizeof *c);
  if (len < 4 * BLOB_HEADER_WORDS) return -1;
  uint32_t h[BLOB_HEADER_WORDS];
  memcpy(h, blob, sizeof h);
  if (h[0] != BLOB_MAGIC || h[1] != 1) return -2;
  c->d = h[2];
  c->num_wires = h[3];
  c->num_routed = h[4];
  c->num_constants = h[5];
  c->num_selectors = h[6];
  c->num_challenges = h[7];
  c->qdf = h[8];
  c->rate_bits = h[9];
  c->cap_height = h[10];
  c->pow_bits = h[11];
  c->num_queries = h[12];
  c->n_steps = h[13];
  for (int i = 0; i < 8; i++) c->arity_bits[i] = h[14 + i];
  c->hasher = h[22];
  c->num_gates = h[23];
  c->num_pi = h[24];
  c->flags = h[25];
  c->num_pp = h[26];
  memcpy(c->digest_in, &h[32], 32);
  if (c->d > 26 || c->n_steps > 8 || c->hasher > 1 || c->num_challenges > 4) return -3;
  c->n = (size_t)1 << c->d;
  c->N = c->n << c->rate_bits;
  size_t off = 4 * BLOB_HEADER_WORDS;
  if (len < off + (size_t)c->num_gates * 4 * BLOB_GATE_WORDS) return -1;
  c->gates = (gate_t *)calloc(c->num_gates ? c->num_gates : 1, sizeof(gate_t));
  c->num_gate_constraints = 0;
  for (uint32_t i = 0; i < c->num_gates; i++) {
    uint32_t g[BLOB_GATE_WORDS];
    memcpy(g, blob + off, sizeof g);
    off += sizeof g;
    gate_t *G = &c->gates[i];
    G->kind = g[0];
    memcpy(G->p, &g[1], 16);
    G->sel_index = g[5];
    G->group_start = g[6];
    G->group_end = g[7];
    G->num_constraints = g[8];
    G->degree = g[9];
    G->num_constants = g[10];
    if (G->kind >= G_KIND_COUNT) return -4;
    if (G->num_constraints != gate_num_constraints(G->kind, G->p)) return -5;
    if (G->num_constraints > c->num_gate_constraints) c->num_gate_constraints = G->num_constraints;
  }
  if (c->flags & 2) {
    c->cap_in = blob + off;
    off += ((size_t)32) << c->cap_height;
  }
  size_t need = off + 8 * ((size_t)c->num_routed + (size_t)c->num_constants * c->n + (size_t)c->num_routed * c->n);
  if (len < need) return -1;
  if (off % 8) return -6;
  c->k_is = (const gl_t *)(blob + off);
  off += 8 * (size_t)c->num_routed;
  c->constants = (const gl_t *)(blob + off);
  off += 8 * (size_t)c->num_constants * c->n;
  c->sigmas = (const gl_t *)(blob + off);
  return 0;
}

int circuit_load(circuit_t *c, const uint8_t *blob, size_t len) {
  int rc = circuit_parse(c, blob, len);
  if (rc) return rc;
  /* constants_sigmas commitment: columns = [constants... | sigmas...] */
  size_t ncs = (size_t)c->num_constants + c->num_routed;
  gl_t *vals = (gl_t *)malloc(sizeof(gl_t) * ncs * c->n);
  memcpy(vals, c->constants, sizeof(gl_t) * (size_t)c->num_constants * c->n);
  memcpy(vals + (size_t)c->num_constants * c->n, c->sigmas, sizeof(gl_t) * (size_t)c->num_routed * c->n);
  batch_from_values(&c->cs, vals, ncs, c->d, c->rate_bits, c->cap_height);
  free(vals);
  if (c->cap_in) {
    for (size_t i = 0; i < ((size_t)1 << c->cap_height); i++)
      if (memcmp(c->cap_in + 32 * i, c->cs.tree.cap[i].b, DIGEST_BYTES)) return -7;
  }
  if (c->flags & 1) {
    memcpy(c->circuit_digest.b, c->digest_in, DIGEST_BYTES);
  } else {
    /* circuit_builder.rs build(): circuit_digest = H::hash_no_pad(cap.flatten() ||
     * hash_pad(domain_separator = []) .to_vec() || [degree_bits])   (pad to the sponge RATE; pinned by tests/test_reference_proofs.py) */
    size_t ncap = (size_t)1 << c->cap_height;
    gl_t *parts = (gl_t *)malloc(sizeof(gl_t) * (4 * ncap + 5));
    for (size_t i = 0; i < ncap; i++) digest_to_elems(&c->cs.tree.cap[i], parts + 4 * i);
    digest_t ds = kh_hash_pad(NULL, 0);
    digest_to_elems(&ds, parts + 4 * ncap);
    parts[4 * ncap + 4] = c->d;
    c->circuit_digest = kh_hash_no_pad(parts, 4 * ncap + 5);
    free(parts);
  }
  return 0;
}
int circuit_load_verifier(circuit_t *c, const uint8_t *blob, size_t len, const uint8_t *cap, const uint8_t *digest) {
  int rc = circuit_parse(c, blob, len);
  if (rc) return rc;
  size_t ncap = (size_t)1 << c->cap_height;
  c->vcap = (digest_t *)malloc(sizeof(digest_t) * ncap);
  for (size_t i = 0; i < ncap; i++) memcpy(c->vcap[i].b, cap + DIGEST_BYTES * i, DIGEST_BYTES);
  c->cs.tree.cap = c->vcap;
  memcpy(c->circuit_digest.b, digest, DIGEST_BYTES);
  return 0;
}
void circuit_free(circuit_t *c) {
  free(c->gates);
  c->gates = NULL;
  free(c->vcap);
  c->vcap = NULL;
  if (c->cs.coeffs) batch_free(&c->cs);
}
