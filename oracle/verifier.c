/*
 * oracle/verifier.c -- CPU restatement of plonky2 0.2.2 `verifier::verify`
 * (+ fri/verifier.rs) for KeccakGoldilocksConfig, D = 2, operating on the
 * uncompressed proof bytes of SURVEY.md C.11.  TEST INFRASTRUCTURE ONLY.
 *
 * Reference call sites: plonky2-backend/src/actions/verify_action.rs:11-17
 * (verify_compressed) and every in-tree test's
 * `assert!(circuit_data.verify(proof).is_ok())`, e.g.
 * circuit_translation/tests/test_precompiled.rs:43.  Acceptance by this
 * verifier is the system-level invariant the reference's tests check.
 */
#include "oracle.h"
#include "circuit.h"
#include <stdlib.h>
#include <string.h>

const circuit_t *orc_circuit_inner(const orc_circuit *oc);

typedef struct {
  const uint8_t *p;
  size_t len, pos;
  int bad;
} rd_t;
static const uint8_t *rd(rd_t *r, size_t n) {
  if (r->pos + n > r->len) {
    r->bad = 1;
    return NULL;
  }
  const uint8_t *q = r->p + r->pos;
  r->pos += n;
  return q;
}
static uint64_t rd_u64(rd_t *r) {
  const uint8_t *q = rd(r, 8);
  uint64_t v = 0;
  if (q) memcpy(&v, q, 8);
  return v;
}
static ext_t rd_ext(rd_t *r) {
  ext_t e;
  e.c0 = rd_u64(r);
  e.c1 = rd_u64(r);
  if (e.c0 >= GL_P || e.c1 >= GL_P) r->bad = 1;
  return e;
}
static digest_t rd_digest(rd_t *r) {
  digest_t d;
  const uint8_t *q = rd(r, DIGEST_BYTES);
  if (q) memcpy(d.b, q, DIGEST_BYTES);
  else memset(d.b, 0, DIGEST_BYTES);
  return d;
}

static ext_t reduce_ext(const ext_t *v, size_t n, ext_t alpha) {
  ext_t acc = ext_from(0);
  for (size_t i = n; i-- > 0;) acc = ext_add(ext_mul(acc, alpha), v[i]);
  return acc;
}

/* fri/verifier.rs compute_evaluation: interpolate the arity points of the
 * coset through beta. */
static ext_t compute_evaluation(gl_t x, size_t x_in_coset, unsigned ab, const ext_t *evals, ext_t beta) {
  size_t arity = (size_t)1 << ab;
  gl_t g = gl_root_of_unity(ab);
  ext_t ys[64];
  gl_t xs[64];
  size_t rev = bitrev(x_in_coset, ab);
  gl_t start = gl_mul(x, gl_pow(g, arity - rev));
  gl_t p = 1;
  for (size_t i = 0; i < arity; i++) {
    xs[i] = gl_mul(start, p);
    ys[i] = evals[bitrev(i, ab)];
    p = gl_mul(p, g);
  }
  ext_t res = ext_from(0);
  for (size_t i = 0; i < arity; i++) {
    ext_t num = ys[i];
    gl_t den = 1;
    for (size_t j = 0; j < arity; j++) {
      if (j == i) continue;
      num = ext_mul(num, ext_sub(beta, ext_from(xs[j])));
      den = gl_mul(den, gl_sub(xs[i], xs[j]));
    }
    res = ext_add(res, ext_scale(num, gl_inv(den)));
  }
  return res;
}

int orc_verify(const orc_circuit *oc, const uint8_t *proof, size_t len, orc_trace *tr) {
  const circuit_t *c = orc_circuit_inner(oc);
  g_hasher = c->hasher == 1;
  const size_t n = c->n, N = c->N;
  const unsigned d = c->d, rb = c->rate_bits, lgN = d + rb, chh = c->cap_height;
  const size_t W = c->num_wires, R = c->num_routed, K = c->num_challenges, QF = c->qdf, NC = c->num_constants;
  const size_t nchunks = (R + QF - 1) / QF, PP = nchunks - 1;
  const size_t ncap = (size_t)1 << chh;
  const size_t nzp = K * (1 + PP), ncs = NC + R, nall = ncs + W + nzp + K * QF;
  orc_trace ltr;
  if (!tr) tr = &ltr;
  memset(tr, 0, sizeof *tr);
  (void)n;

  rd_t r = {proof, len, 0, 0};
  digest_t *caps = (digest_t *)malloc(sizeof(digest_t) * ncap * (3 + c->n_steps));
  for (size_t i = 0; i < 3 * ncap; i++) caps[i] = rd_digest(&r);
  const digest_t *wires_cap = caps, *zs_cap = caps + ncap, *q_cap = caps + 2 * ncap;
  digest_t *step_caps = caps + 3 * ncap;
  /* openings in OpeningSet order */
  ext_t *o_const = (ext_t *)malloc(sizeof(ext_t) * (nall + K));
  ext_t *o_sig = o_const + NC, *o_wires = o_sig + R, *o_zs = o_wires + W, *o_zs_next = o_zs + K;
  ext_t *o_pp = o_zs_next + K, *o_quot = o_pp + K * PP;
  for (size_t i = 0; i < nall + K; i++) o_const[i] = rd_ext(&r);
  for (size_t i = 0; i < c->n_steps * ncap; i++) step_caps[i] = rd_digest(&r);
  /* remember where the queries start; parse the tail first (final poly, pow, PIs) */
  size_t final_len = (N >> rb);
  for (uint32_t s = 0; s < c->n_steps; s++) final_len >>= c->arity_bits[s];
  size_t oracle_cols[4] = {ncs, W, nzp, K * QF};
  size_t qbytes = 0;
  {
    size_t sib0 = lgN - chh;
    for (int o = 0; o < 4; o++) qbytes += 8 * oracle_cols[o] + 1 + DIGEST_BYTES * sib0;
    size_t lg = lgN;
    for (uint32_t s = 0; s < c->n_steps; s++) {
      unsigned ab = c->arity_bits[s];
      lg -= ab;
      qbytes += 16 * ((size_t)1 << ab) + 1 + DIGEST_BYTES * (lg - chh);
    }
  }
  size_t queries_pos = r.pos;
  size_t tail_pos = queries_pos + qbytes * c->num_queries;
  size_t expect_len = tail_pos + 16 * final_len + 8 + 8 * (size_t)c->num_pi;
  int rc = ORC_E_VERIFY;
  ext_t *final_poly = NULL, *b0 = NULL;
  gl_t *pis = NULL;
  gl_t pow_witness = 0;
  if (r.bad || expect_len != len) goto done;
  r.pos = tail_pos;
  final_poly = (ext_t *)malloc(sizeof(ext_t) * (final_len ? final_len : 1));
  for (size_t i = 0; i < final_len; i++) final_poly[i] = rd_ext(&r);
  pow_witness = rd_u64(&r);
  pis = (gl_t *)malloc(sizeof(gl_t) * (c->num_pi ? c->num_pi : 1));
  for (uint32_t i = 0; i < c->num_pi; i++) pis[i] = rd_u64(&r);
  if (r.bad) goto done;

  /* ---- challenges (get_challenges / fri_challenges) ---- */
  gl_t pih[4];
  poseidon_hash_no_pad(pis, c->num_pi, pih);
  challenger_t ch;
  ch_init(&ch);
  ch_observe_digest(&ch, &c->circuit_digest);
  ch_observe_many(&ch, pih, 4);
  ch_observe_cap(&ch, wires_cap, ncap);
  gl_t betas[4], gammas[4], alphas[4];
  for (size_t k = 0; k < K; k++) betas[k] = ch_get(&ch);
  for (size_t k = 0; k < K; k++) gammas[k] = ch_get(&ch);
  ch_observe_cap(&ch, zs_cap, ncap);
  for (size_t k = 0; k < K; k++) alphas[k] = ch_get(&ch);
  ch_observe_cap(&ch, q_cap, ncap);
  ext_t zeta = ch_get_ext(&ch);
  /* to_fri_openings: zeta batch = constants, sigmas, wires, zs, partial products, quotient; then zs_next */
  b0 = (ext_t *)malloc(sizeof(ext_t) * nall);
  {
    size_t t = 0;
    for (size_t i = 0; i < ncs + W + K; i++) b0[t++] = o_const[i];
    for (size_t i = 0; i < K * PP + K * QF; i++) b0[t++] = o_pp[i];
  }
  for (size_t i = 0; i < nall; i++) ch_observe_ext(&ch, b0[i]);
  for (size_t i = 0; i < K; i++) ch_observe_ext(&ch, o_zs_next[i]);
  ext_t alpha = ch_get_ext(&ch);
  ext_t fri_betas[8];
  for (uint32_t s = 0; s < c->n_steps; s++) {
    ch_observe_cap(&ch, step_caps + s * ncap, ncap);
    fri_betas[s] = ch_get_ext(&ch);
    tr->fri_betas[s][0] = fri_betas[s].c0;
    tr->fri_betas[s][1] = fri_betas[s].c1;
  }
  for (size_t i = 0; i < final_len; i++) ch_observe_ext(&ch, final_poly[i]);
  ch_observe(&ch, pow_witness);
  gl_t pow_resp = ch_get(&ch);
  size_t qidx[64];
  for (uint32_t q = 0; q < c->num_queries; q++) {
    qidx[q] = (size_t)(ch_get(&ch) % N);
    tr->query_indices[q] = (uint32_t)qidx[q];
  }
  memcpy(tr->pi_hash, pih, sizeof pih);
  memcpy(tr->betas, betas, sizeof betas);
  memcpy(tr->gammas, gammas, sizeof gammas);
  memcpy(tr->alphas, alphas, sizeof alphas);
  tr->zeta[0] = zeta.c0;
  tr->zeta[1] = zeta.c1;
  tr->alpha_fri[0] = alpha.c0;
  tr->alpha_fri[1] = alpha.c1;
  tr->pow_witness = pow_witness;

  /* ---- plonk identity at zeta (verifier.rs verify_with_challenges) ---- */
  {
    ext_t *terms = (ext_t *)malloc(sizeof(ext_t) * (K + K * nchunks + 3 * c->num_gate_constraints + 8));
    ext_t pih_e[4];
    for (int i = 0; i < 4; i++) pih_e[i] = ext_from(pih[i]);
    ext_t zn = zeta;
    for (unsigned i = 0; i < d; i++) zn = ext_mul(zn, zn);
    ext_t z_h = ext_sub(zn, ext_from(1));
    /* eval_l_0(n, x) = (x^n - 1) / (n (x - 1)) */
    ext_t l0 = ext_mul(z_h, ext_inv(ext_scale(ext_sub(zeta, ext_from(1)), (gl_t)c->n)));
    size_t t = 0;
    for (size_t k = 0; k < K; k++) terms[t++] = ext_mul(l0, ext_sub(o_zs[k], ext_from(1)));
    for (size_t k = 0; k < K; k++) {
      for (size_t m = 0; m < nchunks; m++) {
        ext_t prev = m == 0 ? o_zs[k] : o_pp[k * PP + m - 1];
        ext_t next = m == nchunks - 1 ? o_zs_next[k] : o_pp[k * PP + m];
        ext_t np = ext_from(1), dp = ext_from(1);
        for (size_t j = m * QF; j < (m + 1) * QF && j < R; j++) {
          ext_t s_id = ext_scale(zeta, c->k_is[j]);
          np = ext_mul(np, ext_add(ext_add(o_wires[j], ext_scale(s_id, betas[k])), ext_from(gammas[k])));
          dp = ext_mul(dp, ext_add(ext_add(o_wires[j], ext_scale(o_sig[j], betas[k])), ext_from(gammas[k])));
        }
        terms[t++] = ext_sub(ext_mul(prev, np), ext_mul(next, dp));
      }
    }
    eval_gate_constraints_ext(c, o_const, o_wires, pih_e, terms + t, terms + t + c->num_gate_constraints);
    t += c->num_gate_constraints;
    int ok = 1;
    for (size_t k = 0; k < K; k++) {
      ext_t van = reduce_ext(terms, t, ext_from(alphas[k]));
      ext_t qz = reduce_ext(o_quot + k * QF, QF, zn);
      if (!ext_eq(van, ext_mul(z_h, qz))) ok = 0;
    }
    free(terms);
    if (!ok) goto done;
  }

  /* ---- FRI (fri/verifier.rs verify_fri_proof) ---- */
  if (c->pow_bits && (pow_resp >> (64 - c->pow_bits)) != 0) goto done;
  {
    ext_t red0 = reduce_ext(b0, nall, alpha);
    ext_t red1 = reduce_ext(o_zs_next, K, alpha);
    ext_t g_zeta = ext_scale(zeta, gl_root_of_unity(d));
    ext_t aK = ext_pow(alpha, K);
    gl_t wN = gl_root_of_unity(lgN);
    ext_t *ev = (ext_t *)malloc(sizeof(ext_t) * nall);
    r.pos = queries_pos;
    for (uint32_t q = 0; q < c->num_queries; q++) {
      size_t x = qidx[q];
      size_t t = 0;
      const gl_t *zrow = NULL;
      for (int o = 0; o < 4; o++) {
        const uint8_t *leafp = rd(&r, 8 * oracle_cols[o]);
        const uint8_t *lp = rd(&r, 1);
        if (!leafp || !lp || *lp != lgN - chh) goto fri_fail;
        digest_t sib[64];
        for (unsigned i = 0; i < *lp; i++) sib[i] = rd_digest(&r);
        gl_t *leaf = (gl_t *)malloc(8 * oracle_cols[o]);
        memcpy(leaf, leafp, 8 * oracle_cols[o]);
        const digest_t *cap = o == 0 ? c->cs.tree.cap : o == 1 ? wires_cap : o == 2 ? zs_cap : q_cap;
        int ok = merkle_verify(leaf, oracle_cols[o], x, cap, chh, sib, *lp);
        for (size_t j = 0; j < oracle_cols[o]; j++) {
          if (leaf[j] >= GL_P) ok = 0;
          ev[t++] = ext_from(leaf[j]);
        }
        (void)zrow;
        free(leaf);
        if (!ok) goto fri_fail;
      }
      /* fri_combine_initial */
      gl_t sx = gl_mul(GL_GENERATOR, gl_pow(wN, bitrev(x, lgN)));
      ext_t sxe = ext_from(sx);
      ext_t e0 = reduce_ext(ev, nall, alpha);
      ext_t e1 = reduce_ext(ev + ncs + W, K, alpha);
      ext_t sum = ext_mul(ext_sub(e0, red0), ext_inv(ext_sub(sxe, zeta)));
      sum = ext_add(ext_mul(sum, aK), ext_mul(ext_sub(e1, red1), ext_inv(ext_sub(sxe, g_zeta))));
      ext_t old_eval = sum;
      size_t lg = lgN;
      for (uint32_t s = 0; s < c->n_steps; s++) {
        unsigned ab = c->arity_bits[s];
        size_t arity = (size_t)1 << ab;
        ext_t evals[64];
        gl_t flat[128];
        for (size_t i = 0; i < arity; i++) {
          evals[i] = rd_ext(&r);
          flat[2 * i] = evals[i].c0;
          flat[2 * i + 1] = evals[i].c1;
        }
        const uint8_t *lp = rd(&r, 1);
        lg -= ab;
        if (!lp || *lp != lg - chh) goto fri_fail;
        digest_t sib[64];
        for (unsigned i = 0; i < *lp; i++) sib[i] = rd_digest(&r);
        size_t coset_index = x >> ab, within = x & (arity - 1);
        if (!ext_eq(evals[within], old_eval)) goto fri_fail;
        old_eval = compute_evaluation(sx, within, ab, evals, fri_betas[s]);
        if (!merkle_verify(flat, 2 * arity, coset_index, step_caps + s * ncap, chh, sib, *lp)) goto fri_fail;
        for (unsigned i = 0; i < ab; i++) sx = gl_sqr(sx);
        x = coset_index;
      }
      /* final_poly.eval(subgroup_x) == old_eval */
      ext_t fe = ext_from(0);
      for (size_t i = final_len; i-- > 0;) fe = ext_add(ext_scale(fe, sx), final_poly[i]);
      if (!ext_eq(fe, old_eval)) goto fri_fail;
      if (r.bad) goto fri_fail;
    }
    free(ev);
    rc = ORC_OK;
    goto done;
  fri_fail:
    free(ev);
    rc = ORC_E_VERIFY;
  }
done:
  free(b0);
  free(caps);
  free(o_const);
  free(final_poly);
  free(pis);
  return rc;
}
